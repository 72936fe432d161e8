#!/usr/bin/env python
"""What hot USERS cost: BASELINE configs[1] sizes with Zipf(alpha) user activity (duplicate (u, i)
pairs dropped like synth_triples: a user holds at most I interactions), the staged step through the C epoch loop, and the
length of the longest user run of a batch.  Under rocprofv3 --kernel-trace --stats (tools/kstats.sh) the edge launch
shows the chain walk:   python tools/hot_user_probe.py [alpha=0.8] [B=2097152]
(alpha must leave the most active user fewer than I interactions - 1.0 does not at these sizes: no negative is left
for it and the sampler raises like the reference's)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from daisyrec_amd import ops  # noqa: E402

alpha = float(sys.argv[1]) if len(sys.argv) > 1 else 0.8
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 21
U, I, nnz, d = 1_000_000, 100_000, 50_000_000, 64
dev = torch.device("cuda")
g = torch.Generator(device=dev)
g.manual_seed(2022)
if alpha > 0:
    w = 1.0 / torch.arange(1, U + 1, device=dev, dtype=torch.float64) ** alpha
    cdf = torch.cumsum(w / w.sum(), 0)
    u = torch.searchsorted(cdf, torch.rand(nnz, device=dev, generator=g, dtype=torch.float64)).clamp_(max=U - 1)
else:
    u = torch.randint(0, U, (nnz,), device=dev, generator=g, dtype=torch.int64)
i = torch.randint(0, I, (nnz,), device=dev, generator=g, dtype=torch.int64)
key = torch.unique((u << 32) | i)
del u, i
users, items = (key >> 32).to(torch.int32), (key & 0xFFFFFFFF).to(torch.int32)
n = users.numel()
indptr, csr = ops.build_user_csr(users, items, U)
triples = torch.stack([users, items, torch.zeros_like(items)], 1).contiguous()
ops.resample_neg_per_interaction(indptr, csr, I, triples, 2022, 0)
top = int(torch.bincount(users.long(), minlength=U).max())
del users, items, key
Q = torch.empty(I, d, device=dev).normal_(0.0, 0.01, generator=g)
P = torch.empty(U, d, device=dev).normal_(0.0, 0.01, generator=g)
ctx = ops.BprContext(B, d, U, I, device=dev)
index = ops.TrainIndex(triples, U, I, user_sorted=True)
plan = ops.EpochPlan(n, U, I, device=dev).build_indexed(index, B, order="feistel", seed=1, epoch=0)
nb = plan.num_batches
ub = plan.read_batch(0, B)[0]
longest = int(torch.bincount(ub.long()).max())


def epoch():
    ctx.fit_epoch_sgd(plan, P, Q, 1e-4, 1e-3, 1e-3, item_mode=ops.ITEM_MODES["fused"])


epoch()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
a.record()
epoch()
b.record()
torch.cuda.synchronize()
ms = a.elapsed_time(b) / nb
print(f"users Zipf({alpha}): {n} interactions, the most active user holds {top} ({100.0 * top / n:.2f} %); batch 0: its longest "
      f"user run {longest} samples;  B={B}: {ms * 1e3:.1f} us per step = {1548 * B / (ms * 1e-3) / 8e12:.3f} of the 8 TB/s roof "
      f"by the 1548-B model (plan build not timed)", flush=True)
