"""Summarise gpurun_out/prof_<tag>/ (tools/collect_profiles.sh) into profiles/<tag>/:
  kernel_stats.csv   rocprofv3 --stats per-kernel table (as written by rocprofv3)
  pmc_traffic.json   per kernel: launches, average duration (us), FETCH_SIZE / WRITE_SIZE per launch in bytes (RAW counter x 1024: the factors of
                     profiles/<round>/counter_calibration.json -- FETCH_SIZE x 2.0, WRITE_SIZE x 1.0 at every access width, measured by tools/calibrate_counters.sh --
                     are applied by tools/roofline_defs.py when it forms a stage's traffic; Infinity-Cache hits are included) and the SQ counters of the same step
  kernel_stats_f4.csv  rocprofv3 --stats of the TIMED configuration (four frames per step on four streams)
  bench.json         the bench line of the same build without the profiler
  roofline.json      the bench line's `roofline` object recomputed from THESE files alone (tools/roofline_defs.py: stage time = summed rocprofv3 durations of
                     its kernels per frame; issued lane-operations = SQ_INSTS_VALU x 64; traffic = FETCH_SIZE + WRITE_SIZE) -- bench.py's live object uses
                     the same functions with HIP-event stage times"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(ROOT, "profiles", tag)
os.makedirs(dst, exist_ok=True)


def short(name):
    n = name.replace("void (anonymous namespace)::", "")
    return n.split("(")[0]


def pmc(dirname):
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(src, dirname, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            out[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return out


def durations(dirname):
    out = collections.defaultdict(list)
    for f in glob.glob(os.path.join(src, dirname, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            out[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return out


for f in glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True):
    shutil.copy(f, os.path.join(dst, "kernel_stats.csv"))
for f in glob.glob(os.path.join(src, "stats_f4", "**", "*kernel_stats.csv"), recursive=True):
    shutil.copy(f, os.path.join(dst, "kernel_stats_f4.csv"))   # the timed F = 4 step (kernels of four frames overlap: durations include the sharing)
dur = durations("stats")
fetch, write, sq = pmc("pmc_fetch"), pmc("pmc_write"), pmc("pmc_sq")
res = {}
for k in sorted(dur, key=lambda k: -sum(dur[k])):
    if "elementwise" in k or "Fill" in k or "copy" in k.lower():
        continue
    e = {"launches": len(dur[k]), "avg_us": sum(dur[k]) / len(dur[k])}
    if k in fetch and fetch[k].get("FETCH_SIZE"):
        v = fetch[k]["FETCH_SIZE"]; e["fetch_bytes_per_launch"] = sum(v) / len(v) * 1024
    if k in write and write[k].get("WRITE_SIZE"):
        v = write[k]["WRITE_SIZE"]; e["write_bytes_per_launch"] = sum(v) / len(v) * 1024
    if k in sq:
        e["sq"] = {c: sum(v) / len(v) for c, v in sq[k].items()}
    res[k] = e
json.dump(res, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)
if os.path.exists(os.path.join(src, "bench.json")):
    shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, "bench.json"))
for k, e in list(res.items())[:16]:
    print(f"{k[:44]:44s} n={e['launches']:3d} avg={e['avg_us']:8.1f}us fetch={e.get('fetch_bytes_per_launch', 0)/1e6:8.2f}MB write={e.get('write_bytes_per_launch', 0)/1e6:8.2f}MB")
if os.path.exists(os.path.join(src, "extra_kernels.txt")):
    shutil.copy(os.path.join(src, "extra_kernels.txt"), os.path.join(dst, "extra_kernels.txt"))
for f in glob.glob(os.path.join(src, "extra_stats", "**", "*kernel_stats.csv"), recursive=True):
    shutil.copy(f, os.path.join(dst, "extra_kernel_stats.csv"))

# ---- the roofline object from the profile alone
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import roofline_defs as rd  # noqa: E402
frames = rd.frames_of(res)
if frames:
    n_sb = 2040   # the profiled workload: 3840 x 2160
    stage_ms = {}
    for st in rd.ALG_BYTES_PER_SB:
        us, _, _, _ = rd.stage_counters(res, st, frames)
        if us > 0 and st not in ("fwd_txfm_quant", "inv_txfm_recon"):
            stage_ms[st] = us * 1e-3
    roof = rd.roofline(stage_ms, n_sb, res, f"profiles/{tag}/pmc_traffic.json")
    json.dump(roof, open(os.path.join(dst, "roofline.json"), "w"), indent=1)
    print("roofline:", {k: roof[k] for k in ("stage", "bound", "achieved", "peak", "frac", "useful_frac", "traffic_over_algorithmic")})
    for st, e in roof["stages"].items():
        print(f"  {st:22s} {e['ms']:7.3f} ms  alg {e['algorithmic_GBps']:7.0f} GB/s  traffic/alg {e.get('traffic_over_algorithmic', float('nan')):6.1f}  issued {e.get('issued_frac', float('nan')):5.2f}  useful {e.get('useful_frac', float('nan')):5.2f}")
