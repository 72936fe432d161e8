#!/usr/bin/env python
"""Turn rocprofv3 PMC passes into per-step HBM traffic for bench.py's roofline.traffic.

Usage (on the GPU box, separate passes as MI355X_MICROARCH.md prescribes: FETCH_SIZE costs 3 of
the 4 TCC slots, WRITE_SIZE 2, so they cannot share a pass; no trace domains beside --kernel-trace):
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_rd -o rd -- python bench.py --steps 8 --warmup 2 --no-cpu-baseline
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_wr -o wr -- python bench.py --steps 8 --warmup 2 --no-cpu-baseline
    python tools/pmc_traffic.py gpurun_out/pmc_rd gpurun_out/pmc_wr profiles/pmc_traffic.json

Corrections (MI355X_MICROARCH.md §HBM): FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports
exactly half of the bytes of wide (16 B/lane) coalesced reads, which is what the row gathers are, so
the read side is doubled; WRITE_SIZE is taken as reported (uncalibrated).  Infinity-Cache hits are
counted by these fabric-side counters, so this is traffic leaving L2, an upper bound of HBM bytes.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

STEP_KERNELS = ("k_staged_user_edges", "k_staged_user", "k_staged_item_edges", "k_staged_item", "k_item_apply_counts",
                "k_fwd", "k_reduce_partials", "k_finalize", "k_item_grad_chunked", "k_item_edges", "k_item_grad_sorted",
                "k_item_grad_atomic", "k_item_reg", "k_user_chunked", "k_user_edges", "k_user", "k_item_apply",
                "k_unorm_reduce", "k_unorm")
PLAN_KERNELS = ("k_part_count", "k_part_scatter", "k_invert_perm", "k_plan_keys", "k_plan_entries", "k_run_finish")


def load(dirpath, counter):
    files = glob.glob(os.path.join(dirpath, "**", "*counter_collection.csv"), recursive=True)
    per_kernel = defaultdict(lambda: [0, 0.0])
    for f in files:
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != counter:
                continue
            name = row["Kernel_Name"]
            per_kernel[name][0] += 1
            per_kernel[name][1] += float(row["Counter_Value"])
    return per_kernel


def short(name, kernels=STEP_KERNELS):
    for k in kernels:
        if k + "<" in name or name.startswith("daisy::" + k) or ("::" + k + "(") in name or ("::" + k + "<") in name:
            return k
    return None


def main():
    rd_dir, wr_dir, out = sys.argv[1:4]
    batch = int(sys.argv[4]) if len(sys.argv) > 4 else 2097152     # interactions per step of the profiled runs (bench.py default)
    rd, wr = load(rd_dir, "FETCH_SIZE"), load(wr_dir, "WRITE_SIZE")
    table = {}
    steps = None
    for src, key, corr in ((rd, "read", 2.0), (wr, "write", 1.0)):
        for name, (cnt, kib) in src.items():
            k = short(name)
            if k is None:
                continue
            t = table.setdefault(k, {"launches": cnt, "read_bytes_per_launch": 0.0, "write_bytes_per_launch": 0.0})
            t[f"{key}_bytes_per_launch"] += kib * 1024.0 * corr / max(cnt, 1)
            if k in ("k_fwd", "k_staged_user"):
                steps = cnt
    per_step = sum(v["read_bytes_per_launch"] + v["write_bytes_per_launch"] for v in table.values())
    # the epoch plan builds of the same run (bytes per build, all kernels of a build summed)
    plan = {}
    for src, key, corr in ((rd, "read", 2.0), (wr, "write", 1.0)):
        for name, (cnt, kib) in src.items():
            k = short(name, PLAN_KERNELS)
            if k is not None:
                t = plan.setdefault(k, {"launches": cnt, "read_bytes": 0.0, "write_bytes": 0.0})
                t[f"{key}_bytes"] += kib * 1024.0 * corr
    source = sys.argv[5] if len(sys.argv) > 5 else "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py (see profiles/)"
    res = {"hbm_bytes_per_step": per_step, "batch_per_step": batch, "hbm_bytes_per_interaction": per_step / batch,
           "steps_profiled": steps, "source": source, "per_kernel": table, "plan_kernels_total_bytes": plan,
           "corrections": "FETCH_SIZE KiB x1024 x2 (gfx950 wide-read undercount), WRITE_SIZE KiB x1024 x1; "
                          "fabric-side counters: Infinity-Cache hits included (upper bound of HBM bytes)"}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
