#!/bin/bash
# Run ON THE GPU BOX: wall time of the reference encoder and of the hooked one (all hooks) on the same synthetic clips -> gpurun_out/enc_wall/
# Four applications (oracle/Makefile.enc): ref = unpatched reference, C kernels; hip = hooks on the C build; simd = unpatched reference as its x86 build
# dispatches it (SSE2 .. AVX-512); hip_simd = hooks on the SIMD build.  The hooked ones run twice (the second run has warm module loads).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/enc_wall
mkdir -p $OUT
cd $R
PRESET=${PRESET:-6}
LP=${LP:-8}
IFS=, read -ra GEO_LIST <<< "${GEOS:-1280 720 8,1920 1080 8}"   # GEOS="3840 2160 4" for other sizes
APPS=${APPS:-"ref hip hip simd hip_simd hip_simd"}
for geo in "${GEO_LIST[@]}"; do
  set -- $geo; W=$1; H=$2; N=$3
  python - $W $H $N ${BD:-8} <<'PY'
import sys; sys.path.insert(0, "tests")
import e2e_common as E
w, h, n, bd = map(int, sys.argv[1:5])
E.make_clip("gpurun_out/enc_wall/clip.yuv", w, h, n, seed=3, bd=bd)
PY
  ARGS="-i $OUT/clip.yuv -w $W -h $H -n $N --preset $PRESET --fps 30 -q 36 --lp $LP"
  [ "${BD:-8}" = 10 ] && ARGS="$ARGS --input-depth 10"   # BD=10: 10-bit clips (the 16-bit pipeline of the hooks)
  for app in $APPS; do
    # NAME_res = application NAME with SVT_HIP_RESIDENT=1 (source-side planes stay on the device between their writes, integration/svt_hip_hooks.c),
    # e.g. APPS="simd hip_simd hip_simd hip_simd_res hip_simd_res"
    bin=${app%_res}; res=0; [ "$bin" != "$app" ] && res=1
    [ -x $R/oracle/_ref/SvtAv1EncApp_$bin ] || { echo "${W}x${H} $app: not built" | tee -a $OUT/wall.txt; continue; }
    s=$(date +%s.%N)
    case $app in
      ref|simd) timeout 900 $R/oracle/_ref/SvtAv1EncApp_$bin $ARGS -b $OUT/$app.ivf > $OUT/${app}_$W.log 2>&1 ;;
      *) SVT_HIP_RESIDENT=$res SVT_HIP_HOOKS=${HOOKS:-all} timeout 900 $R/oracle/_ref/SvtAv1EncApp_$bin $ARGS -b $OUT/$app.ivf > $OUT/${app}_$W.log 2>&1 ;;
    esac
    e=$(date +%s.%N)
    log=$OUT/${app}_$W.log
    echo "${W}x${H} n=$N preset=$PRESET lp=$LP $app wall_s=$(python -c "print(round($e - $s, 2))") $(grep -h 'Total Encoding Time\|Average Speed' $log | tr -s '\t\n' '  ')" | tee -a $OUT/wall.txt
  done
  base=ref; [ -f $OUT/ref.ivf ] || base=simd   # the unpatched application of this run (the SIMD build codes the C build's bitstream)
  for app in hip simd hip_simd hip_res hip_simd_res; do
    [ "$app" != "$base" ] && [ -f $OUT/$app.ivf ] && { [ -f $OUT/$base.ivf ] || { echo "${W}x${H} $app: no unpatched run in this invocation to compare with"; continue; }; cmp -s $OUT/$base.ivf $OUT/$app.ivf && echo "${W}x${H} $app bitstream identical to $base" || echo "${W}x${H} $app BITSTREAM DIFFERS from $base"; } | tee -a $OUT/wall.txt
  done
  for app in hip_res hip_simd_res; do
    [ -f $OUT/${app}_$W.log ] && grep -h "svt_hip_resident\|svt_hip_context\|svt_hip_lf_pictures\|svt_hip_warmup\|svt_hip_alloc_cache\|svt_hip_md_pre" $OUT/${app}_$W.log | sed "s/^/$app /" | tee -a $OUT/wall.txt
    [ -f $OUT/${app}_$W.log ] && grep -h "svt_hip_hook_time" $OUT/${app}_$W.log | awk -v a=$app '{printf "%s %s %s %s | ", a, $2, $3, $4} END {print ""}' | tee -a $OUT/wall.txt
  done
  for app in hip hip_simd hip_res hip_simd_res; do
    [ -f $OUT/${app}_$W.log ] && grep -h "svt_hip_hook " $OUT/${app}_$W.log | awk -v a=$app '{h+=substr($3,9); f+=substr($4,10)} END {print a, "hook launches:", h, "fallbacks:", f}' | tee -a $OUT/wall.txt
  done
  rm -f $OUT/clip.yuv $OUT/*.ivf
done
