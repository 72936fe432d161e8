#!/usr/bin/env python
"""Wall time of the legs `bench.py --gpus 8` runs, measured on ONE GPU (what a rank, and rank 0's single-GPU reference, do):
the data builds at a rank's share of BASELINE configs[2] (62.5 M interactions) and at the full set (500 M), the measure()
calls of the main point, the sweep and the same-global-batch reference.  DESIGN.md section 5 quotes the output."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device("cuda")
a = argparse.Namespace(dist="uniform", plan="auto", item_mode="fused", nnz=None, reg=0.001, overlap_plan=0)


def t(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = fn()
    torch.cuda.synchronize()
    return r, time.perf_counter() - t0


for world in (8, 1):
    data, s = t(lambda: bench.build_data(a, 0, world, dev, "c3"))
    print(f"build_data(c3, world={world}): {s:.1f} s  (n = {data['n']}, index {data['index'].nbytes / 1e9:.1f} GB)", flush=True)
    if world == 8:
        for B, steps in ((1 << 24, 25), (1 << 21, 18), (1 << 23, 18)):
            _, s = t(lambda: bench.measure(a, data, 0, 1, dev, B, 1, steps, 0))
            print(f"  measure(B_local={B}, {steps} steps, single rank, no exchange): {s:.1f} s", flush=True)
    else:
        for B in (1 << 21, 1 << 24, 1 << 26, 1 << 27):
            try:
                r, s = t(lambda: bench.measure(a, data, 0, 1, dev, B, 1, None, 4))
                print(f"  one-GPU reference point B={B}: {s:.1f} s ({r['steps']} steps, {r['dt'] / r['steps'] * 1e3:.2f} ms per step)", flush=True)
            except Exception as e:  # noqa: BLE001
                print(f"  one-GPU reference point B={B}: {type(e).__name__}: {e}", flush=True)
                torch.cuda.empty_cache()
    bench.free_data(data)
