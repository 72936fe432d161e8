#!/usr/bin/env python
"""Soak run: many epochs of the staged step at the C2 shapes (plan rebuilt every epoch, device shuffle), watching the
epoch loss, the table norms and the device memory: python tools/soak.py [epochs]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from daisyrec_amd import ops  # noqa: E402

dev = torch.device("cuda")
U, I, nnz, d, B = 1_000_000, 100_000, 50_000_000, 64, 1 << 21
epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 30
triples = bench.synth_triples(U, I, nnz, 2022, dev)
n = triples.shape[0]
g = torch.Generator(device=dev)
g.manual_seed(1)
Q = torch.empty(I, d, device=dev).normal_(0.0, 0.01, generator=g)
P = torch.empty(U, d, device=dev).normal_(0.0, 0.01, generator=g)
index = ops.TrainIndex(triples, U, I, user_sorted=True)
plan = ops.EpochPlan(n, U, I, device=dev)
ctx = ops.BprContext(B, d, U, I, device=dev)
mem0 = None
t0 = time.perf_counter()
for ep in range(epochs):
    plan.build_indexed(index, B, order="feistel", seed=7, epoch=ep)
    ctx.epoch_acc.zero_()
    ctx.fit_epoch_sgd(plan, P, Q, 0.05, 1e-3, 1e-3, item_mode=ops.ITEM_MODES["fused"])
    acc = ctx.epoch_acc.cpu()
    mem = torch.cuda.memory_allocated()
    mem0 = mem if mem0 is None else mem0
    if ep % 5 == 0 or ep == epochs - 1:
        print(f"epoch {ep:3d}  loss/interaction {float(acc[0]) / n:.6f}  non-finite steps {int(acc[1])}  "
              f"|P| {float(P.norm()):.3f}  |Q| {float(Q.norm()):.3f}  torch memory {mem / 2**30:.3f} GiB "
              f"(+{(mem - mem0) / 2**20:.1f} MiB)", flush=True)
    assert int(acc[1]) == 0 and mem == mem0
torch.cuda.synchronize()
print(f"{epochs} epochs x {n} interactions in {time.perf_counter() - t0:.2f} s "
      f"({epochs * n / (time.perf_counter() - t0) / 1e9:.2f} G interactions/s including plan builds and the per-epoch host sync)")
