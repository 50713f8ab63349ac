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
for i in 1 2 3; do run device_$i ""; run host_$i "SVT_HIP_DLF_EDGES=host"; done
run device_verbose "SVT_HIP_VERBOSE=1"; run host_verbose "SVT_HIP_VERBOSE=1 SVT_HIP_DLF_EDGES=host"
for v in device host; do
  echo "$v: $(grep -h 'dlf_search: picture up' $OUT/${v}_verbose.log | awk '{u+=$5; m+=$11; e+=$14; p+=$17; n++} END {printf "level search per picture: up %.2f, mode info %.2f, edges %.2f, probes %.2f ms (n=%d)", u/n, m/n, e/n, p/n, n}')" | tee -a $OUT/ab.txt
  echo "$v: $(grep -h 'dlf: levels' $OUT/${v}_verbose.log | sed 's/.*edges \([0-9.]*\) ms.*/\1/' | awk '{e+=$1; n++} END {if (n) printf "filter, edges per picture: %.2f ms (n=%d)", e/n, n; else print "no picture was filtered"}')" | tee -a $OUT/ab.txt
done
for f in device_1 host_1 device_verbose host_verbose; do cmp -s $OUT/warm.ivf $OUT/$f.ivf && echo "$f bitstream identical to the first run" || echo "$f BITSTREAM DIFFERS"; done | tee -a $OUT/ab.txt
(cd $R/gpurun_out/dlf_edges && rm -f clip.yuv warm.ivf device_?.ivf host_?.ivf device_verbose.ivf host_verbose.ivf)
