cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c14
for v in base nt base nt; do
  cp tools/ab/lib_$v.so svt-av1_amd/libsvtav1_hip.so
  timeout 600 python bench.py --no-sweep --no-variants --no-1080p --no-cpu-baseline --no-transfers --steps 30 > gpurun_out/c14/bench_$v.json 2> gpurun_out/c14/bench_$v.err
  python - $v <<'PY'
import json,sys
v=sys.argv[1]
try:
    d=json.loads(open(f'gpurun_out/c14/bench_{v}.json').read().strip().splitlines()[-1])
    st=d['config']['stages_ms']
    print(v, "value", round(d['value']), "ms/step", round(d['ms_per_step'],3), "sgr_search", round(st['sgr_units_search'],4), "parity", d['config']['parity_spot_check'])
except Exception as e:
    print(v, "failed", e, open(f'gpurun_out/c14/bench_{v}.err').read()[-800:])
PY
done | tee gpurun_out/c14/ab.txt
