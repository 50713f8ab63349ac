cd $GRAFT_REPO_ROOT; python tools/_dbg.py 2>&1 | tail -12; timeout 900 python -m pytest tests/test_md_pre_gpu.py -q -x -m gpu 2>&1 | tail -5
