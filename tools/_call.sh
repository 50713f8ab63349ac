cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c5
timeout 1200 python -m pytest tests/test_sgr_gpu.py tests/test_wiener_gpu.py -q -x -m gpu 2>&1 | tail -3 | tee gpurun_out/c5/pytest_a.txt
timeout 1200 python -m pytest tests/test_encode_e2e.py -q -x -m gpu -k "hooked_encode_on_gpu or wiener" 2>&1 | tail -3 | tee gpurun_out/c5/pytest_b.txt
