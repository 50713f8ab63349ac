#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repo root: rocprofv3 kernel-trace statistics of the default bench step and the two
# HBM-traffic PMC passes (FETCH_SIZE and WRITE_SIZE do not fit one pass; PMC passes carry no API trace domains).
# Output: gpurun_out/prof_<tag>/{stats,pmc_fetch,pmc_write}/..., summarised by tools/summarize_profiles.py into profiles/<tag>/.
set -u
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# one frame per step on one stream, so that a kernel's trace duration is its own (the default bench overlaps the kernels of several frames on forked streams)
BENCH="python $R/bench.py --steps 12 --warmup 2 --frames 1 --groups 4 --no-sweep --no-transfers --no-cpu-baseline --no-variants"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o k -- $BENCH > $OUT/stats.log 2>&1
# ... and the TIMED configuration itself (four frames per step on four streams: a kernel's duration here includes what it shares the chip with) -> kernel_stats_f4.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_f4 -o k -- python $R/bench.py --steps 12 --warmup 2 --no-sweep --no-transfers --no-cpu-baseline --no-variants --no-1080p > $OUT/stats_f4.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o p -- $BENCH > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o p -- $BENCH > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT/pmc_sq -o p -- $BENCH > $OUT/pmc_sq.log 2>&1
cd $R && PYTHONFAULTHANDLER=1 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench.py rc=$? bytes=$(wc -c < $OUT/bench.json)" >> $OUT/bench.err
[ -s $OUT/bench.json ] || { sleep 2; python bench.py > $OUT/bench.json 2>> $OUT/bench.err; echo "bench.py (second attempt) rc=$? bytes=$(wc -c < $OUT/bench.json)" >> $OUT/bench.err; }
# kernels outside the bench step: HIP-event timings
(python tools/wiener_time.py; python tools/tf_time.py; python tools/tf_time.py --bd 10; python tools/compound_time.py; python tools/hbd_time.py --bd 8; python tools/hbd_time.py --bd 10; python tools/cdef_pick_time.py; SVT_HIP_CDEF_SELECT=resident python tools/cdef_pick_time.py; PICK_MAG=31 python tools/cdef_pick_time.py; PICK_MAG=31 SVT_HIP_CDEF_SELECT=resident python tools/cdef_pick_time.py) > $OUT/extra_kernels.txt 2>&1
# ... and their rocprofv3 kernel statistics (one trace over the three timing tools)
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/extra_stats -o x -- python $R/tools/extra_time.py > $OUT/extra_stats.log 2>&1
cd $R
# ENCODER=1: the hooked encoder as well -- kernel + memory-copy trace of a 12-frame 4K encode (copy share, tools/copy_trace_summary.py) and the loop-filter hooks' sub-step times
if [ "${ENCODER:-0}" = 1 ]; then
  N=12 bash tools/encoder_copy_trace.sh > $OUT/encoder_copy_trace.log 2>&1
  python tools/copy_trace_summary.py > $OUT/copy_trace_4k.txt 2>&1
  cp gpurun_out/enc_copy/stats/k_kernel_stats.csv $OUT/encoder_hooks_4k_kernel_stats.csv 2>/dev/null
  (cd gpurun_out/enc_copy/stats && rm -f k_kernel_trace.csv k_memory_copy_trace.csv)
  bash tools/dlf_edges_ab.sh > $OUT/dlf_edges_ab.log 2>&1; cp gpurun_out/dlf_edges/ab.txt $OUT/dlf_edges_ab.txt 2>/dev/null
fi
ls -R $OUT | head -40
