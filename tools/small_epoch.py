#!/usr/bin/env python
"""us per step at the reference's default batch (B=256, ml-100k shapes): persistent-workgroup epoch vs the
per-step phase kernels vs the staged step."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from daisyrec_amd import ops  # noqa: E402

U, I, n, B = 943, 1152, 78363, int(sys.argv[1]) if len(sys.argv) > 1 else 256
d = int(sys.argv[2]) if len(sys.argv) > 2 else 32
g = torch.Generator().manual_seed(0)
tri = torch.stack([torch.randint(0, U, (n,), generator=g), torch.randint(0, I, (n,), generator=g),
                   torch.randint(0, I, (n,), generator=g)], 1).to(torch.int32).cuda()
P, Q = torch.randn(U, d, device="cuda") * 0.01, torch.randn(I, d, device="cuda") * 0.01
ctx = ops.BprContext(B, d, U, I)
plan = ops.EpochPlan(n, U, I).build(tri, B, order="feistel", seed=1, epoch=0)
index = ops.TrainIndex(tri, U, I)
plan_i = ops.EpochPlan(n, U, I).build_indexed(index, B, order="feistel", seed=1, epoch=0)
nb = plan.num_batches
for name, pl, mode in (("one workgroup per epoch", plan, "fused"), ("phase kernels (sorted)", plan, "sorted"),
                       ("phase kernels (atomic)", plan, "atomic"), ("staged step, partitioned plan", plan_i, "fused")):
    for _ in range(2):
        ctx.fit_epoch_sgd(pl, P, Q, 0.01, 1e-3, 1e-3, item_mode=ops.ITEM_MODES[mode])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        ctx.fit_epoch_sgd(pl, P, Q, 0.01, 1e-3, 1e-3, item_mode=ops.ITEM_MODES[mode])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print(f"B={B} d={d} {name:32s} {1e6 * dt / nb:7.2f} us/step  {1e3 * dt:7.2f} ms/epoch  {n / dt / 1e6:7.2f} M interactions/s")
