#!/bin/bash
# Run ON THE GPU BOX: the four-frame step with one stage left out at a time -- what each stage costs INSIDE the overlapped step (not alone on an idle chip).
# Stages that consume a left-out stage's output run on stale buffers: times only, no parity.
ALL="pyr hme me subpel enc_txfm dlf cdef_search cdef_pick cdef_apply sgr_units sgr_apply"
run() { python bench.py --steps 20 --warmup 3 --no-sweep --no-transfers --no-cpu-baseline --no-variants --no-1080p --stages "$1" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f' % d['ms_per_step'])"; }
full=$(echo $ALL | tr ' ' ',')
echo "all: $(run $full) ms per step"
for x in $ALL; do
  s=$(echo $ALL | tr ' ' '\n' | grep -vx $x | tr '\n' ',' | sed 's/,$//')
  echo "without $x: $(run $s) ms per step"
done
