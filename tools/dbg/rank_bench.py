"""MF.rank at scale: B users x C candidates (MFRecommender.py:106-123) - scores + per-user stable descending sort + top-k."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from daisyrec_amd import ops
U, I, d, C, topk = 1_000_000, 100_000, 64, 1000, 50
g = torch.Generator(device="cuda"); g.manual_seed(0)
P = torch.randn(U, d, device="cuda", generator=g) * 0.01
Q = torch.randn(I, d, device="cuda", generator=g) * 0.01
for B in (128, 8192, 65536):
    us = torch.randint(0, U, (B,), device="cuda", generator=g)
    cands = torch.randint(0, I, (B, C), device="cuda", generator=g)
    ops.mf_rank_topk(P, Q, us, cands, topk)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        ops.mf_rank_topk(P, Q, us, cands, topk)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    print(f"rank_topk B={B} C={C} topk={topk}: {ms:.3f} ms  = {B / ms / 1e3:.2f} M users/s, {B * C / ms / 1e6:.2f} G scores/s")
