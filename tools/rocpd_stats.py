#!/usr/bin/env python3
"""Per-kernel statistics out of a rocprofv3 (1.1, rocpd sqlite output) --kernel-trace database:
    python tools/rocpd_stats.py gpurun_out/prof/x_results.db [substring ...] -> name, calls, avg/min/max us, total ms
Used to produce the summaries committed under profiles/."""
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
    ks = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
    cols = [r[1] for r in cur.execute(f"pragma table_info({ks})")]
    namecol = "display_name" if "display_name" in cols else ("kernel_name" if "kernel_name" in cols else cols[-1])
    rows = cur.execute(f"select s.{namecol}, count(*), avg(d.end - d.start), min(d.end - d.start), max(d.end - d.start), sum(d.end - d.start), "
                       f"max(d.grid_size_x * d.grid_size_y * d.grid_size_z / (d.workgroup_size_x * d.workgroup_size_y * d.workgroup_size_z)) "
                       f"from {kd} d join {ks} s on d.kernel_id = s.id group by s.{namecol} order by 6 desc").fetchall()
    want = sys.argv[2:]
    print("kernel,calls,avg_us,min_us,max_us,total_ms,max_workgroups")
    for name, n, avg, mn, mx, tot, wgs in rows:
        short = name
        short = short.replace("(anonymous namespace)::", "")
        if short.startswith("void "): short = short[5:]
        short = short.split("(")[0]
        if want and not any(w in short for w in want):
            continue
        print(f"{short},{n},{avg / 1e3:.2f},{mn / 1e3:.2f},{mx / 1e3:.2f},{tot / 1e6:.3f},{wgs}")


if __name__ == "__main__":
    main()
