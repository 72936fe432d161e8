#!/usr/bin/env python
"""Where a launch-bound step spends its time: from a rocprofv3 kernel trace (`--kernel-trace --output-format csv`) of
tools/sweep_batch.py, per batch size: the kernels of one step (average duration each), the sum of their durations, the
idle time between consecutive kernels (start[i+1] - end[i]) and the span per step.  If span ~ sum + gaps of 1.5-2 us the
chain is bound by the device's kernel boundaries; gaps of 5+ us mean the host cannot launch fast enough.

    rocprofv3 --kernel-trace --output-format csv -d DIR -o t -- python tools/sweep_batch.py 1024 16384 65536
    python tools/trace_gaps.py DIR
"""
import csv
import glob
import os
import sys
from collections import defaultdict

STEP = ("k_staged_user<", "k_staged_user_edges<", "k_staged_item<", "k_staged_item_edges<", "k_unorm", "k_bpr_small",
        "k_staged_step_small", "k_staged_adam")


def short(name):
    name = name.replace("daisy::", "")
    for cut in ("<", "("):
        if cut in name:
            name = name[:name.index(cut)]
    return name.replace("void ", "")


def main():
    src = sys.argv[1]
    tr = glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)
    rows = []
    for row in csv.DictReader(open(tr[0])):
        rows.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]), row["Kernel_Name"],
                     int(row.get("Grid_Size_X", row.get("Grid_Size", 0)) or 0)))
    rows.sort()
    # split the trace into runs of step kernels with the same grid signature (one run per batch size of the sweep)
    runs, cur, sig = [], [], None
    for r in rows:
        n = r[2]
        if not any(k in n for k in STEP):
            if len(cur) > 40:
                runs.append(cur)
            cur, sig = [], None
            continue
        cur.append(r)
    if len(cur) > 40:
        runs.append(cur)
    for run in runs:
        # steps = occurrences of the user pass
        nsteps = sum(1 for r in run if "k_staged_user<" in r[2]) or 1
        dur = defaultdict(lambda: [0, 0])
        gaps = []
        for a, b in zip(run, run[1:]):
            gaps.append(b[0] - a[1])
        for r in run:
            d = dur[short(r[2])]
            d[0] += 1; d[1] += r[1] - r[0]
        span = run[-1][1] - run[0][0]
        ksum = sum(v[1] for v in dur.values())
        gaps.sort()
        gx = max(r[3] for r in run if "k_staged_user<" in r[2]) if any("k_staged_user<" in r[2] for r in run) else 0
        print(f"-- {nsteps} steps, user-pass grid {gx}: span {span / nsteps / 1e3:.2f} us/step, kernels {ksum / nsteps / 1e3:.2f} us/step, "
              f"gaps {sum(gaps) / nsteps / 1e3:.2f} us/step (median gap {gaps[len(gaps) // 2] / 1e3:.2f} us, "
              f"p90 {gaps[int(len(gaps) * 0.9)] / 1e3:.2f} us), {len(run) / nsteps:.1f} launches/step")
        for k, (c, t) in sorted(dur.items(), key=lambda kv: -kv[1][1]):
            print(f"     {k:28s} {c / nsteps:5.2f} per step  {t / c / 1e3:8.2f} us each")


if __name__ == "__main__":
    main()
