#!/bin/bash
# Run ON THE GPU BOX: the restoration unit search per frame chain against ONE pair of launches for the step's four frames
run() { python bench.py --steps 30 --warmup 4 --no-sweep --no-transfers --no-cpu-baseline --no-variants --no-1080p 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1  %.3f ms per step  %.0f SB/s  parity %s' % (d['ms_per_step'], d['value'], d['config']['parity_spot_check']))"; }
run "per frame "
SVT_BENCH_SGR_JOINT=1 run "joint     "
run "per frame "
SVT_BENCH_SGR_JOINT=1 run "joint     "
