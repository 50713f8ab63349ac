#!/bin/bash
# Run ON THE GPU BOX: the step at several frames-per-step values (ms per frame), same box, back to back
for f in ${FRAMES:-2 3 4 5 6 8}; do
  python bench.py --frames $f --steps 20 --warmup 3 --no-sweep --no-transfers --no-cpu-baseline --no-variants --no-1080p 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('F=$f  %.3f ms per step  %.3f ms per frame  %.0f SB/s' % (d['ms_per_step'], d['ms_per_step']/$f, d['value']))"
done
