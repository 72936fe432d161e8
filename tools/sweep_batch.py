#!/usr/bin/env python
"""Whole epochs through daisy_bpr_fit_epoch_sgd (the loop MF.fit runs, in C) at the C2 shapes for several batch sizes:
python tools/sweep_batch.py [B ...]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from daisyrec_amd import ops  # noqa: E402

dev = torch.device("cuda")
U, I, nnz, d = 1_000_000, 100_000, 50_000_000, 64
triples = bench.synth_triples(U, I, nnz, 2022, dev)
n_all = triples.shape[0]
g = torch.Generator(device=dev)
g.manual_seed(1)
Q = torch.empty(I, d, device=dev).normal_(0.0, 0.01, generator=g)
P = torch.empty(U, d, device=dev).normal_(0.0, 0.01, generator=g)
for B in [int(x) for x in sys.argv[1:]] or [1024, 4096, 16384, 65536, 262144, 1 << 20, 1 << 21]:
    n = min(n_all, B * 2000) // B * B            # at most 2000 steps per epoch
    tr = triples[:n].contiguous()
    index = ops.TrainIndex(tr, U, I, user_sorted=True)
    plan = ops.EpochPlan(n, U, I, device=dev).build_indexed(index, B, order="feistel", seed=1, epoch=0)
    ctx = ops.BprContext(B, d, U, I, device=dev)
    ctx.epoch_acc.zero_()
    ctx.fit_epoch_sgd(plan, P, Q, 0.01, 1e-3, 1e-3, item_mode=ops.ITEM_MODES["fused"])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ctx.fit_epoch_sgd(plan, P, Q, 0.01, 1e-3, 1e-3, item_mode=ops.ITEM_MODES["fused"])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    nb = n // B
    us = dt / nb * 1e6
    print(f"B={B:8d}  {nb:5d} steps  {us:9.2f} us/step  {n / dt / 1e9:.3f} G inter/s  "
          f"{1548 * n / dt / 8e12:.3f} of 8 TB/s", flush=True)
    ctx.close(); plan.close(); index.close()
