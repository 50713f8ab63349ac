#!/bin/bash
# Run ON THE GPU BOX: fork / join per step against free-running frame slots
run() { python bench.py --steps 40 --warmup 4 --no-sweep --no-transfers --no-cpu-baseline --no-variants --no-1080p 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1  %.3f ms per step  %.0f SB/s  parity %s' % (d['ms_per_step'], d['value'], d['config']['parity_spot_check']))"; }
run "fork/join      "
SVT_BENCH_FREERUN=1 run "free run       "
SVT_BENCH_FREERUN=1 SVT_BENCH_FREERUN_LAG_US=0 run "free run, lag 0"
SVT_BENCH_FREERUN=1 SVT_BENCH_FREERUN_LAG_US=900 run "free run, lag 900 us"
run "fork/join      "
