"""Summarise gpurun_out/enc_copy/stats (tools/encoder_copy_trace.sh): GPU time by kernel class, and the 2-D copy kernels attributed to the hook stage they belong to
(the nearest non-copy kernels before / after them on the same stream)."""
import collections
import csv
import sys

d = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/enc_copy/stats"
rows = list(csv.DictReader(open(f"{d}/k_kernel_trace.csv")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))


def short(n):
    return n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:44]


dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
tot = sum(dur(r) for r in rows)
cls = collections.Counter()
for r in rows:
    n = r["Kernel_Name"]
    cls["2-D copies (copyBufferRect)" if "copyBufferRect" in n else "small copies (copyBuffer)" if "copyBuffer" in n else "fills" if "fillBuffer" in n else "kernels"] += dur(r)
mc = list(csv.DictReader(open(f"{d}/k_memory_copy_trace.csv")))
sdma = sum((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in mc)
print(f"kernel-trace time {tot:.1f} ms over {len(rows)} dispatches; DMA-engine copies {sdma:.1f} ms over {len(mc)}")
for k, v in cls.most_common():
    print(f"  {k:32s} {v:8.2f} ms  {100 * v / tot:5.1f} % of kernel-trace time")
by = collections.defaultdict(list)
for r in rows:
    by[(r["Queue_Id"], r["Stream_Id"])].append(r)
att = collections.Counter(); cnt = collections.Counter(); mx = collections.Counter()
for seq in by.values():
    for i, r in enumerate(seq):
        if "copyBufferRect" not in r["Kernel_Name"]:
            continue
        nxt = next((short(q["Kernel_Name"]) for q in seq[i + 1:] if "rocclr" not in q["Kernel_Name"]), "(end)")
        prv = next((short(q["Kernel_Name"]) for q in reversed(seq[:i]) if "rocclr" not in q["Kernel_Name"]), "(start)")
        k = (prv, nxt)
        att[k] += dur(r); cnt[k] += 1; mx[k] = max(mx[k], dur(r))
print("2-D copies by neighbourhood (after -> before), ms total / count / longest:")
for k, v in att.most_common(10):
    print(f"  {v:7.2f} {cnt[k]:4d} {mx[k]:6.2f}   {k[0]} -> {k[1]}")
top = collections.Counter()
for r in rows:
    top[short(r["Kernel_Name"])] += dur(r)
print("top kernels:")
for k, v in top.most_common(8):
    print(f"  {v:8.2f} ms {100 * v / tot:5.1f} %  {k}")
