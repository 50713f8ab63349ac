#!/bin/bash
# Run ON THE GPU BOX: rocprofv3 kernel + memory-copy trace of the hooked SIMD encoder on a 6-frame 4K clip -> gpurun_out/enc_copy/stats (which copies are slow, by size and direction)
set -u
export N=${N:-6}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/enc_copy
mkdir -p $OUT
cd $R && python - <<'PY'
import sys; sys.path.insert(0, "tests")
import e2e_common as E
E.make_clip("gpurun_out/enc_copy/clip.yuv", 3840, 2160, int(__import__("os").environ.get("N", "6")), seed=3, bd=8)
PY
cd /tmp && export TMPDIR=/tmp
SVT_HIP_HOOKS=${HOOKS:-all} rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $OUT/stats -o k -- $R/oracle/_ref/SvtAv1EncApp_hip_simd -i $OUT/clip.yuv -w 3840 -h 2160 -n ${N:-6} --preset 6 --fps 30 -q 36 --lp 8 -b $OUT/o.ivf > $OUT/enc.log 2>&1
cd $OUT && rm -f clip.yuv o.ivf
ls -la $OUT/stats | head; ls $OUT/stats/* | head -20
