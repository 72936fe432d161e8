#!/usr/bin/env python
"""Staged step at the C2 shapes for several factor counts d (the reference's mf.yaml default is 100):
python tools/sweep_factors.py [d ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from daisyrec_amd import ops  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from probe_staged import ev_time  # noqa: E402

dev = torch.device("cuda")
U, I, nnz, B = 1_000_000, 100_000, 50_000_000, 1 << 21
triples = bench.synth_triples(U, I, nnz, 2022, dev)
n = triples.shape[0]
index = ops.TrainIndex(triples, U, I, user_sorted=True)
MODE = os.environ.get("MODE", "fused")
plan = ops.EpochPlan(n, U, I, device=dev)
if MODE == "fused":
    plan.build_indexed(index, B, order="feistel", seed=1, epoch=0)
else:
    plan.build(triples, B, order="feistel", seed=1, epoch=0, user_sorted=True)
from daisyrec_amd.model.MFRecommender import padded_factors  # noqa: E402
PITCH = os.environ.get("PITCH", "auto")          # what MF / FM train on (config['row_pitch']); PITCH=0: the bare row shapes
for d in [int(x) for x in sys.argv[1:]] or [64, 100, 128, 96, 50, 32, 24, 200, 256]:
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    dp = padded_factors(d, 0 if PITCH == "0" else PITCH)
    Q = torch.zeros(I, dp, device=dev)
    P = torch.zeros(U, dp, device=dev)
    Q[:, :d].normal_(0.0, 0.01, generator=g)
    P[:, :d].normal_(0.0, 0.01, generator=g)
    ctx = ops.BprContext(B, dp, U, I, device=dev)
    k = [0]

    def step():
        ctx.set_batch_from_plan(plan, k[0] % 20)
        ctx.sgd_step(P, Q, 0.01, 1e-3, 1e-3, item_mode=ops.ITEM_MODES[MODE])
        k[0] += 1

    for _ in range(3):
        step()
    ms = ev_time(step, 20)
    alg = 24 * d + 12            # bytes per interaction: 6 row touches of 4d bytes + the (u, i, j) record
    print(f"d={d:4d}{'' if dp == d else f' (rows padded to {dp})':20s}  {ms:.4f} ms/step  {B / ms / 1e6:.3f} G inter/s  {alg * B / (ms * 1e-3) / 1e12:.2f} TB/s algorithmic "
          f"({alg * B / (ms * 1e-3) / 8e12:.3f} of 8 TB/s)", flush=True)
    ctx.close()
