cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c13
timeout 1200 python -m pytest tests/test_encode_e2e.py -q -x -m gpu -k "md_pre" 2>&1 | tail -4 | tee gpurun_out/c13/pytest.txt
for geo in "1920 1080 16" "1280 720 16"; do
  for hooks in all all,md_pre; do
    GEOS="$geo" HOOKS=$hooks APPS="simd hip_simd_res hip_simd_res hip_simd_res" PRESET=6 LP=8 bash tools/encoder_walltime.sh > gpurun_out/c13/wall_${hooks//,/_}_${geo// /x}.log 2>&1
    grep -h "wall_s\|identical\|DIFFERS\|svt_hip_md_pre" gpurun_out/c13/wall_${hooks//,/_}_${geo// /x}.log | sort -u | sed "s/^/[$hooks] /"
  done
done | tee gpurun_out/c13/summary.txt
