#!/bin/bash
# Run ON THE GPU BOX: wall time of the unpatched reference encoder and of the hooked one (all hooks) on the same synthetic clips -> gpurun_out/enc_wall/
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/enc_wall
mkdir -p $OUT
cd $R
IFS=, read -ra GEO_LIST <<< "${GEOS:-1280 720 8,1920 1080 8}"   # GEOS="3840 2160 4" for other sizes
for geo in "${GEO_LIST[@]}"; do
  set -- $geo; W=$1; H=$2; N=$3
  python - $W $H $N <<'PY'
import sys; sys.path.insert(0, "tests")
import e2e_common as E
w, h, n = map(int, sys.argv[1:4])
E.make_clip("gpurun_out/enc_wall/clip.yuv", w, h, n, seed=3, bd=8)
PY
  ARGS="-i $OUT/clip.yuv -w $W -h $H -n $N --preset 6 --fps 30 -q 36 --lp 8"
  for app in ref hip hip; do
    s=$(date +%s.%N)
    if [ $app = ref ]; then timeout 600 $R/oracle/_ref/SvtAv1EncApp_ref $ARGS -b $OUT/ref.ivf > $OUT/ref_$W.log 2>&1
    else SVT_HIP_HOOKS=all timeout 600 $R/oracle/_ref/SvtAv1EncApp_hip $ARGS -b $OUT/hip.ivf > $OUT/hip_$W.log 2>&1; fi
    e=$(date +%s.%N)
    log=$OUT/${app}_$W.log
    echo "${W}x${H} n=$N $app wall_s=$(python -c "print(round($e - $s, 2))") $(grep -h 'Total Encoding Time\|Average Speed' $log | tr -s '\t\n' '  ')" | tee -a $OUT/wall.txt
  done
  cmp $OUT/ref.ivf $OUT/hip.ivf && echo "${W}x${H} bitstreams identical" | tee -a $OUT/wall.txt
  grep -h "svt_hip_hook" $OUT/hip_$W.log | awk '{f+=substr($4,10)} END {print "fallbacks:", f}' | tee -a $OUT/wall.txt
  rm -f $OUT/clip.yuv $OUT/ref.ivf $OUT/hip.ivf
done
