#!/bin/bash
# Run ON THE GPU BOX: FETCH_SIZE / WRITE_SIZE of kernels that move a known number of bytes at one access width per lane -> gpurun_out/counter_cal/calibration.json
# (tools/ubench/counter_cal.hip; copy the JSON to profiles/<round>/counter_calibration.json, tools/roofline_defs.py reads the latest one).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/counter_cal; mkdir -p $O
[ -x $R/tools/ubench/counter_cal ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $R/tools/ubench/counter_cal.hip -o $R/tools/ubench/counter_cal
cd /tmp && export TMPDIR=/tmp
$R/tools/ubench/counter_cal > $O/true.txt
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o p -- $R/tools/ubench/counter_cal > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o p -- $R/tools/ubench/counter_cal > $O/write.log 2>&1
python3 - <<PY
import csv, glob, json, collections, re
true = {}
for l in open("$O/true.txt"):
    m = re.match(r"TRUE (\S+<[^>]+>) (\d+)", l)
    if m: true[m.group(1)] = int(m.group(2))
def counters(d, name):
    out = collections.defaultdict(list)
    for f in glob.glob("$O/%s/**/*counter_collection.csv" % d, recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name:
                k = r["Kernel_Name"].replace("void ", "").split("(")[0]
                out[k].append(float(r["Counter_Value"]) * 1024)
    return {k: sum(v) / len(v) for k, v in out.items()}
fe, wr = counters("fetch", "FETCH_SIZE"), counters("write", "WRITE_SIZE")
width = {"unsigned char": 1, "unsigned short": 2, "unsigned int": 4, "uint2": 8, "uint4": 16, "HIP_vector_type<unsigned int, 2u>": 8, "HIP_vector_type<unsigned int, 4u>": 16}
cal = {"read": {}, "write": {}, "raw": {}}
for k, t in true.items():
    kind = "read" if "cal_read" in k else "write"
    w = width[k[k.index("<") + 1:-1]]
    got = None
    for kk, v in (fe if kind == "read" else wr).items():
        if kk.startswith("cal_" + kind) and width.get(kk[kk.index("<") + 1:kk.rindex(">")].strip()) == w: got = v
    cal["raw"][k] = {"true_bytes": t, "counter_bytes": got, "other_direction_counter_bytes": None}
    if got: cal[kind][str(w)] = t / got
json.dump(cal, open("$O/calibration.json", "w"), indent=1)
print(json.dumps({k: cal[k] for k in ("read", "write")}, indent=1))
PY
