#!/bin/bash
# Run ON THE GPU BOX: rocprofv3 kernel statistics of the hooked reference encoder (all hooks) on a synthetic 720p clip -> gpurun_out/enc_prof/
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/enc_prof
W=${W:-1280}; H=${H:-720}; N=${N:-6}   # W=3840 H=2160 N=3 for the 4K profile
mkdir -p $OUT
export W H N
cd $R && python - <<'PY'
import sys; sys.path.insert(0, "tests")
import e2e_common as E
import os
E.make_clip("gpurun_out/enc_prof/clip.yuv", int(os.environ.get("W", 1280)), int(os.environ.get("H", 720)), int(os.environ.get("N", 6)), seed=3, bd=8)
PY
cd /tmp && export TMPDIR=/tmp
SVT_HIP_HOOKS=all rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o k -- $R/oracle/_ref/SvtAv1EncApp_hip -i $OUT/clip.yuv -w $W -h $H -n $N --preset 6 --fps 30 -q 36 --lp 8 -b $OUT/o.ivf > $OUT/enc.log 2>&1
rm -f $OUT/clip.yuv $OUT/o.ivf
grep "svt_hip_hook" $OUT/enc.log | head -20
head -30 $OUT/stats/k_kernel_stats.csv | cut -c1-160
