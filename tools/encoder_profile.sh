#!/bin/bash
# Run ON THE GPU BOX: rocprofv3 kernel statistics of the hooked reference encoder (all hooks) on a synthetic 720p clip -> gpurun_out/enc_prof/
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/enc_prof
mkdir -p $OUT
cd $R && python - <<'PY'
import sys; sys.path.insert(0, "tests")
import e2e_common as E
E.make_clip("gpurun_out/enc_prof/clip.yuv", 1280, 720, 6, seed=3, bd=8)
PY
cd /tmp && export TMPDIR=/tmp
SVT_HIP_HOOKS=all rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o k -- $R/oracle/_ref/SvtAv1EncApp_hip -i $OUT/clip.yuv -w 1280 -h 720 -n 6 --preset 6 --fps 30 -q 36 --lp 8 -b $OUT/o.ivf > $OUT/enc.log 2>&1
rm -f $OUT/clip.yuv $OUT/o.ivf
grep "svt_hip_hook" $OUT/enc.log | head -20
head -30 $OUT/stats/k_kernel_stats.csv | cut -c1-160
