#!/bin/bash
# Run ON THE GPU BOX (gpurun): hook "md_pre" — parity tests, then the wall clock of the hooked SIMD encoder with and without it (identical bitstreams required).
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
OUT=$R/gpurun_out/md_pre
mkdir -p $OUT
timeout 900 python -m pytest tests/test_md_pre_gpu.py tests/test_encode_e2e.py -q -x -m gpu -k "md_pre" 2>&1 | tail -15 > $OUT/pytest.txt
cat $OUT/pytest.txt
for geo in "1920 1080 16" "1280 720 16"; do
  for hooks in all all,md_pre; do
    GEOS="$geo" HOOKS=$hooks APPS="simd hip_simd_res hip_simd_res" PRESET=6 LP=${LP:-8} bash tools/encoder_walltime.sh > $OUT/wall_${hooks//,/_}_${geo// /x}.log 2>&1
    grep -h "wall_s\|identical\|DIFFERS\|svt_hip_md_pre" $OUT/wall_${hooks//,/_}_${geo// /x}.log gpurun_out/enc_wall/hip_simd_res_*.log 2>/dev/null | sort -u | sed "s/^/[$hooks] /"
  done
done | tee $OUT/summary.txt
