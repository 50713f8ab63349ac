#!/bin/bash
# Run ON THE GPU BOX: the PCIe-inclusive step with uploads / downloads / both / neither (same streams and events, copies left out), and enqueue-order / stream variants
run() { python bench.py --steps 20 --warmup 3 --no-sweep --no-cpu-baseline --no-variants --no-1080p 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1  resident %.3f ms  with transfers %.3f ms per step' % (d['ms_per_step'], d['value_with_transfers']['ms_per_step']))"; }
for m in ${MODES:-both}; do
  SVT_BENCH_XFER=$m run "xfer=$m default"
  SVT_BENCH_XFER=$m SVT_BENCH_XFER_ORDER=early run "xfer=$m early"
  SVT_BENCH_XFER=$m SVT_BENCH_XFER_STREAMS=2 run "xfer=$m 2streams"
  SVT_BENCH_XFER=$m SVT_BENCH_XFER_ORDER=early SVT_BENCH_XFER_STREAMS=2 run "xfer=$m early 2streams"
done
