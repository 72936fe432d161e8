#!/usr/bin/env python
"""Per-kernel summary (calls, total, average, min, max, share) from rocprofv3 output: either the
`*_kernel_stats.csv` of `--stats --output-format csv` or a rocpd `*_results.db`."""
import csv
import glob
import os
import sqlite3
import sys


def from_db(path):
    cur = sqlite3.connect(path).cursor()
    return cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                       "from kernels group by name order by 3 desc").fetchall()


def from_trace_csv(path):
    agg = {}
    for row in csv.DictReader(open(path)):
        d = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
        a = agg.setdefault(row["Kernel_Name"], [0, 0, 1 << 62, 0])
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    return sorted(((k, v[0], v[1], v[1] / v[0], v[2], v[3]) for k, v in agg.items()), key=lambda r: -r[2])


def main():
    src = sys.argv[1]
    rows = None
    if os.path.isdir(src):
        dbs = glob.glob(os.path.join(src, "**", "*results.db"), recursive=True)
        tr = glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)
        rows = from_trace_csv(tr[0]) if tr else from_db(dbs[0])
    elif src.endswith(".db"):
        rows = from_db(src)
    else:
        rows = from_trace_csv(src)
    tot = sum(r[2] for r in rows)
    print(f"# rocprofv3 --kernel-trace summary of {src}; total kernel time {tot / 1e6:.3f} ms")
    print(f"{'kernel':100s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'%':>6s}")
    for name, n, t, avg, mn, mx in rows[:40]:
        print(f"{name[:100]:100s} {n:6d} {t / 1e6:10.3f} {avg / 1e3:10.2f} {mn / 1e3:9.2f} {mx / 1e3:9.2f} {100 * t / tot:6.2f}")


if __name__ == "__main__":
    main()
