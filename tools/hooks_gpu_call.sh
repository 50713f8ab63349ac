#!/bin/bash
# Run ON THE GPU BOX: the round-4 hooks work in one call -> gpurun_out/hooks/
#   1. GPU identity tests of the deferred / resident paths   2. encode time against the SIMD reference (tools/encoder_walltime.sh)   3. kernel statistics of a hooked 4K encode
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/hooks
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_encode_e2e.py -m gpu -q -n 4 -k "${TESTS:-deferred or resident or test_hooked_encode_on_gpu or padded or 128 or 1080p or option_variants}" > $OUT/tests.log 2>&1; echo "hooks GPU tests: rc=$?" | tee $OUT/summary.txt
tail -5 $OUT/tests.log | tee -a $OUT/summary.txt
APPS="${APPS:-simd hip_simd hip_simd hip_simd_res hip_simd_res}" GEOS="${GEOS:-1280 720 8,1920 1080 8,3840 2160 4}" timeout 900 bash tools/encoder_walltime.sh > $OUT/walltime.log 2>&1
cp $R/gpurun_out/enc_wall/wall.txt $OUT/wall.txt 2>/dev/null; cat $OUT/wall.txt | tee -a $OUT/summary.txt
W=3840 H=2160 N=3 SVT_HIP_RESIDENT=1 timeout 600 bash tools/encoder_profile.sh > $OUT/profile.log 2>&1
f=$R/gpurun_out/enc_prof/stats/k_kernel_stats.csv
[ -f $f ] && { cp $f $OUT/encoder_hooks_4k_kernel_stats.csv; grep -h "svt_hip_resident\|svt_hip_context\|svt_hip_lf_pictures\|svt_hip_warmup\|svt_hip_hook_time" $R/gpurun_out/enc_prof/enc.log | tee -a $OUT/summary.txt; head -8 $f | cut -c1-160 | tee -a $OUT/summary.txt; }
