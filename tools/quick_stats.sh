#!/bin/bash
# rocprofv3 kernel statistics of one frame per step (run on the GPU box): gpurun_out/quick_stats/k_kernel_stats.csv
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/quick_stats; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o k -- python $R/bench.py --steps 12 --warmup 2 --frames 1 --groups 4 --no-sweep --no-transfers --no-cpu-baseline --no-variants > $OUT/log.txt 2>&1
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/k_kernel_stats.csv")))
for r in rows[:14]: print(r['Name'][:90], r['Calls'], r['AverageNs'], r['MinNs'], r['MaxNs'], r['Percentage'])
PY
