#!/bin/bash
# Run ON THE GPU BOX: the hooked SIMD encoder on one synthetic clip with the deblocking edge planes built on the device (default: one upload of the mode-info grid per
# picture, shared by the level search and the filter) and with SVT_HIP_DLF_EDGES=host (host builder + upload of its output in both stages), interleaved.
# -> gpurun_out/dlf_edges/ab.txt: wall clock, thread time inside the two hooks, the verbose log's per-picture sub-step times of one run each.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/dlf_edges
mkdir -p $OUT
cd $R
W=${W:-3840}; H=${H:-2160}; N=${N:-12}; PRESET=${PRESET:-6}
python - $W $H $N <<'PY'
import sys; sys.path.insert(0, "tests")
import e2e_common as E
w, h, n = map(int, sys.argv[1:4])
E.make_clip("gpurun_out/dlf_edges/clip.yuv", w, h, n, seed=3, bd=8)
PY
ARGS="-i $OUT/clip.yuv -w $W -h $H -n $N --preset $PRESET --fps 30 -q ${Q:-36} --lp 8"
run() {   # name, extra env
  s=$(date +%s.%N)
  env $2 SVT_HIP_HOOKS=all timeout 600 $R/oracle/_ref/SvtAv1EncApp_hip_simd $ARGS -b $OUT/$1.ivf > $OUT/$1.log 2>&1
  e=$(date +%s.%N)
  echo "$1 wall_s=$(python -c "print(round($e - $s, 3))") $(grep -h 'Average Speed' $OUT/$1.log | tr -s '\t\n' '  ') | $(grep -h 'svt_hip_hook_time dlf' $OUT/$1.log | tr '\n' ' ')" | tee -a $OUT/ab.txt
}
run warm ""
for i in 1 2; do run device_$i ""; run host_$i "SVT_HIP_DLF_EDGES=host"; done
run device_verbose "SVT_HIP_VERBOSE=1"; run host_verbose "SVT_HIP_VERBOSE=1 SVT_HIP_DLF_EDGES=host"
for v in device host; do
  python - $OUT/${v}_verbose.log $v <<'PY' | tee -a $OUT/ab.txt
import re, sys
t = open(sys.argv[1], errors="replace").read()
rows = [tuple(map(float, m)) for m in re.findall(r"dlf_search: picture up ([0-9.]+) ms, mode info \(host\) ([0-9.]+) ms, edges ([0-9.]+) ms, probes ([0-9.]+) ms", t)]
if rows:
    n = len(rows); avg = [sum(r[i] for r in rows) / n for i in range(4)]
    print(f"{sys.argv[2]}: level search per picture (n={n}): picture up {avg[0]:.2f}, mode info {avg[1]:.2f}, edges {avg[2]:.2f}, probes {avg[3]:.2f} ms")
f = [float(x) for x in re.findall(r"dlf: levels [^\n]*edges ([0-9.]+) ms", t)]
print(f"{sys.argv[2]}: filter, edges per picture: {sum(f) / len(f):.2f} ms (n={len(f)})" if f else f"{sys.argv[2]}: no picture was filtered")
PY
done
for f in device_1 host_1 device_verbose host_verbose; do cmp -s $OUT/warm.ivf $OUT/$f.ivf && echo "$f bitstream identical to the first run" || echo "$f BITSTREAM DIFFERS"; done | tee -a $OUT/ab.txt
(cd $R/gpurun_out/dlf_edges && rm -f clip.yuv warm.ivf device_?.ivf host_?.ivf device_verbose.ivf host_verbose.ivf)
