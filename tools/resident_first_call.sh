#!/bin/bash
# Run ON THE GPU BOX (gpurun --timeout 1500 -- 'bash tools/resident_first_call.sh').  Record of the first GPU call of round 4: what the resident-plane path owed when it was
# written at the end of round 3 (only the CPU test double had run it).  The path is the default now.  -> gpurun_out/resident/
#   1. its GPU tests (bitstream + reconstruction identical to the unpatched encoder with the planes resident; ordinary -m gpu tests since round 4)
#   2. encode time of the SIMD build with the hooks, without and with resident planes, against the SIMD reference (tools/encoder_walltime.sh, NAME_res applications)
#   3. kernel statistics of a hooked 4K encode with resident planes (the copy kernels' share; round 3 without: 42 %, profiles/r03/encoder_hooks_4k_kernel_stats.csv)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/resident
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_encode_e2e.py -m gpu -k "resident and gpu" -q > $OUT/tests.log 2>&1; echo "resident GPU tests: rc=$?" | tee $OUT/summary.txt
tail -3 $OUT/tests.log | tee -a $OUT/summary.txt
# large pictures, where a plane has many readers (40 ME segments per 4K picture): identity again, with the planes' report
SVT_HIP_RESIDENT=1 timeout 600 python tools/e2e_big.py 2160p_8bit_m6 1080p_8bit_m4 2>&1 | tee -a $OUT/summary.txt
APPS="simd hip_simd hip_simd hip_simd_res hip_simd_res" GEOS="1280 720 8,1920 1080 8,3840 2160 4" timeout 900 bash tools/encoder_walltime.sh > $OUT/walltime.log 2>&1
cp $R/gpurun_out/enc_wall/wall.txt $OUT/wall.txt 2>/dev/null; cat $OUT/wall.txt | tee -a $OUT/summary.txt
W=3840 H=2160 N=3 SVT_HIP_RESIDENT=1 timeout 600 bash tools/encoder_profile.sh > $OUT/profile.log 2>&1
f=$R/gpurun_out/enc_prof/stats/k_kernel_stats.csv
[ -f $f ] && { cp $f $OUT/encoder_hooks_4k_resident_kernel_stats.csv; grep -h "svt_hip_resident\|svt_hip_context" $R/gpurun_out/enc_prof/enc.log | tee -a $OUT/summary.txt; head -8 $f | cut -c1-160 | tee -a $OUT/summary.txt; }
