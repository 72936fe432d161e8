#!/usr/bin/env python
"""What the memory system gives random 256-byte row gathers (the item pass's access pattern) as a function of where the
rows are: table size against the 256 MB Infinity Cache, warm (gathered repeatedly) or cold (2 GB of streaming traffic
between the table's last write and the gather, like the user pass puts between a stage row's store and its two loads).
`python tools/mall_probe.py` -> one line per case; profiles/r05_item_pass_counters.txt quotes it."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from daisyrec_amd import ops  # noqa: E402

dev = torch.device("cuda")


def ev(fn, reps=1):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    d, n = 64, 4_194_304                       # entries of a 2 M-sample batch
    out = torch.zeros(8, device=dev)
    flush = torch.zeros(256 << 20, device=dev)     # 1 GB: read + written once = 2 GB of traffic past every cache
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    print("# random 256-B row gathers, 4 194 304 rows per launch (ops.membench what=0); warm = third of three back-to-back "
          "launches; cold = after the table was rewritten (plain stores, random row order) and 2 GB of other traffic")
    for mb in (64, 128, 221, 256, 384, 512, 1024, 2048):
        rows = mb * (1 << 20) // 256
        table = torch.zeros(rows, d, device=dev)
        idx = torch.randint(0, rows, (n,), device=dev, dtype=torch.int32, generator=g)
        # two loads of the same row, far apart, like the positive and the negative entry of a sample: every row twice
        half = torch.randint(0, rows, (n // 2,), device=dev, dtype=torch.int32, generator=g)
        twice = torch.cat([half, half])[torch.randperm(n, device=dev, generator=g)].contiguous()
        wr = torch.randperm(rows, device=dev, generator=g).to(torch.int32)
        for _ in range(2):
            ops.membench(0, table, idx, out)
        warm = ev(lambda: ops.membench(0, table, idx, out), 3)
        cold = []
        for use in (idx, twice):
            t = []
            for _ in range(3):
                ops.membench(2, table, wr, out)        # rewrite every row (random order, plain stores)
                flush.add_(1.0)
                t.append(ev(lambda: ops.membench(0, table, use, out)))
            cold.append(min(t))
        f = lambda ms: f"{n / ms / 1e6:6.2f} G rows/s = {n * 256 / ms / 1e9:5.2f} TB/s ({ms * 1e3:6.1f} us)"   # noqa: E731
        print(f"table {mb:5d} MB  warm {f(warm)}  cold {f(cold[0])}  cold, every row twice {f(cold[1])}", flush=True)
        del table, idx, half, twice, wr
    # the write side of the same pattern: scattered 256-B row stores into a 512 MB buffer (what the user pass does to the stage)
    rows = 512 * (1 << 20) // 256
    table = torch.zeros(rows, d, device=dev)
    wr = torch.randperm(rows, device=dev, generator=g).to(torch.int32)
    ops.membench(2, table, wr, out)
    ms = ev(lambda: ops.membench(2, table, wr, out), 3)
    print(f"scattered row stores, 512 MB: {rows / ms / 1e6:6.2f} G rows/s = {rows * 256 / ms / 1e9:5.2f} TB/s")


if __name__ == "__main__":
    main()
