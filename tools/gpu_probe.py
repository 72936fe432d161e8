#!/usr/bin/env python
"""Memory-system and per-phase probes on one MI355X (used to place the kernels on the
roofline; results are quoted in DESIGN.md).  Run on the GPU box:  python tools/gpu_probe.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from daisyrec_amd import ops  # noqa: E402

dev = "cuda"


def timeit(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def membench():
    d, n = 64, 1 << 22
    out = torch.zeros(8, device=dev)
    print("== random 256-B row access, n = 4M rows per launch ==")
    for rows in (100_000, 1_000_000, 4_000_000, 16_000_000):
        table = torch.zeros(rows, d, device=dev)
        idx = torch.randint(0, rows, (n,), device=dev, dtype=torch.int32)
        for what, nm in ((0, "gather"), (1, "atomic"), (2, "store"), (3, "rmw")):
            ms = timeit(lambda: ops.membench(what, table, idx, out))
            mult = 2 if what == 3 else 1
            print(f"rows={rows:>9} ({rows * 256 / 1e6:7.1f} MB) {nm:7s} {ms:7.3f} ms  "
                  f"{n * 256 * mult / ms / 1e6:8.1f} GB/s  {n / ms / 1e6:6.2f} Grows/s")
        sidx = torch.sort(idx).values
        for what, nm in ((0, "gather-sorted"), (1, "atomic-sorted")):
            ms = timeit(lambda: ops.membench(what, table, sidx, out))
            print(f"rows={rows:>9} {nm:14s} {ms:7.3f} ms  {n * 256 / ms / 1e6:8.1f} GB/s")
        del table


def phases(U=1_000_000, I=100_000, B=1 << 20, d=64, reg=1e-3):
    print(f"== per-phase timing, U={U} I={I} B={B} d={d} reg={reg} ==")
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    P = torch.randn(U, d, device=dev, generator=g) * 0.01
    Q = torch.randn(I, d, device=dev, generator=g) * 0.01
    u = torch.randint(0, U, (B,), device=dev, generator=g, dtype=torch.int32)
    i = torch.randint(0, I, (B,), device=dev, generator=g, dtype=torch.int32)
    j = torch.randint(0, I, (B,), device=dev, generator=g, dtype=torch.int32)
    tri = torch.stack([u, i, j], 1).contiguous()
    perm = torch.randperm(B, device=dev)
    ctx = ops.BprContext(B, d, U, I)
    algo = (24 * d + 12) * B
    res = {}
    res["set_batch(one-batch plan)"] = timeit(lambda: ctx.set_batch_from_triples(tri, idx=perm, B=B))
    ctx.set_batch_from_triples(tri, idx=perm, B=B)
    res["forward(k_fwd+reduce)"] = timeit(lambda: ctx.forward(P, Q))
    ctx.finalize(reg, reg)
    res["finalize"] = timeit(lambda: ctx.finalize(reg, reg, accumulate=False))

    def item_a():
        ctx.item_grad(P, Q, reg, reg, ops.ITEM_MODES["atomic"])
    res["item_grad atomic (reg)"] = timeit(item_a)
    res["item_grad atomic (no reg)"] = timeit(lambda: ctx.item_grad(P, Q, 0.0, 0.0, ops.ITEM_MODES["atomic"]))
    ctx.gQ.zero_()
    res["item_grad sorted"] = timeit(lambda: ctx.item_grad(P, Q, reg, reg, ops.ITEM_MODES["sorted"]))
    ctx.gQ.zero_()
    res["item_grad chunked (reg)"] = timeit(lambda: ctx.item_grad(P, Q, reg, reg, ops.ITEM_MODES["chunked"]))
    ctx.gQ.zero_()
    res["item_grad chunked (no reg)"] = timeit(lambda: ctx.item_grad(P, Q, 0.0, 0.0, ops.ITEM_MODES["chunked"]))
    ctx.gQ.zero_()
    res["user_sgd (lr=0)"] = timeit(lambda: ctx.user_sgd(P, Q, 0.0, reg, reg))
    res["item_apply (touched)"] = timeit(lambda: (ctx.item_grad(P, Q, 0.0, 0.0, 0), ctx.item_sgd_apply(Q, 0.0))) \
        - res["item_grad atomic (no reg)"]
    res["item_apply dense"] = timeit(lambda: ctx.item_sgd_apply(Q, 0.0, dense=True))
    res["sgd_step atomic  (batch preset)"] = timeit(lambda: ctx.sgd_step(P, Q, 1e-9, reg, reg, item_mode=0))
    res["sgd_step sorted  (batch preset)"] = timeit(lambda: ctx.sgd_step(P, Q, 1e-9, reg, reg, item_mode=1))
    res["sgd_step chunked (batch preset)"] = timeit(lambda: ctx.sgd_step(P, Q, 1e-9, reg, reg, item_mode=2))
    res["sgd_step fused   (batch preset)"] = timeit(lambda: ctx.sgd_step(P, Q, 1e-9, reg, reg, item_mode=3))
    for k, v in res.items():
        print(f"{k:34s} {v:8.3f} ms   {B / v / 1e6:8.3f} G inter/s   algo {algo / v / 1e6:8.1f} GB/s ({algo / v / 1e6 / 8000:5.1%} of 8 TB/s)")
    ctx.close()


def plan_cost(U=1_000_000, I=100_000, n=50_000_000, B=1 << 20):
    print(f"== epoch plan build, n={n} B={B} ==")
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    tri = torch.stack([torch.randint(0, U, (n,), device=dev, generator=g, dtype=torch.int32),
                       torch.randint(0, I, (n,), device=dev, generator=g, dtype=torch.int32),
                       torch.randint(0, I, (n,), device=dev, generator=g, dtype=torch.int32)], 1).contiguous()
    plan = ops.EpochPlan(n, U, I)
    print(f"plan bytes {plan.nbytes / 1e9:.2f} GB")
    for order in ("identity", "feistel"):
        t = timeit(lambda: plan.build(tri, B, order=order, seed=1, epoch=0), iters=3, warm=1)
        print(f"plan.build({order:8s})              {t:8.3f} ms  = {t / (n / B):6.3f} ms per {B}-batch")
    tri_s = tri[torch.sort(tri[:, 0].long(), stable=True).indices].contiguous()
    t = timeit(lambda: plan.build(tri_s, B, order="feistel", seed=1, epoch=0, user_sorted=True), iters=3, warm=1)
    print(f"plan.build(feistel, user-sorted)   {t:8.3f} ms  = {t / (n / B):6.3f} ms per {B}-batch")
    t = timeit(lambda: ops.randperm(n, 1, 0), iters=3, warm=1)
    print(f"randperm(50M, philox sort)  {t:8.3f} ms")
    plan.close()


def frontend(U=1_000_000, I=100_000, n=50_000_000):
    """Front-end rows of SURVEY 8f rank 1 at C2 scale: get_ur -> CSR, BasicNegtiveSampler, candidates."""
    print(f"== front end at U={U} I={I} nnz={n} ==")
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    key = torch.unique((torch.randint(0, U, (n,), device=dev, generator=g, dtype=torch.int64) << 32)
                       | torch.randint(0, I, (n,), device=dev, generator=g, dtype=torch.int64))
    users, items = (key >> 32).to(torch.int32), (key & 0xFFFFFFFF).to(torch.int32)
    n = users.numel()
    t = timeit(lambda: ops.build_user_csr(users, items, U), iters=3, warm=1)
    print(f"build_user_csr (get_ur)            {t:8.3f} ms   {n / t / 1e6:8.2f} G pairs/s")
    indptr, csr = ops.build_user_csr(users, items, U)
    for num_ng in (1, 4):
        t = timeit(lambda: ops.sample_neg_per_user(indptr, csr, I, num_ng, 1, 0), iters=5, warm=1)
        print(f"sample_neg_per_user num_ng={num_ng}      {t:8.3f} ms   {U * num_ng / t / 1e6:8.2f} G negatives/s")
    js = ops.sample_neg_per_user(indptr, csr, I, 1, 1, 0)
    t = timeit(lambda: ops.expand_triples(users, items, js), iters=3, warm=1)
    print(f"expand_triples                     {t:8.3f} ms   {n / t / 1e6:8.2f} G triples/s")
    tri = ops.expand_triples(users, items, js)
    t = timeit(lambda: ops.resample_neg_per_interaction(indptr, csr, I, tri, 1, 0), iters=3, warm=1)
    print(f"resample_neg_per_interaction       {t:8.3f} ms   {n / t / 1e6:8.2f} G negatives/s")
    # candidates: 100k test users x 1000 candidates, a 10-item test row per user
    tu = torch.arange(0, 100_000, device=dev, dtype=torch.int64)
    te_items = torch.randint(0, I, (100_000, 10), device=dev, generator=g, dtype=torch.int64)
    te_key = torch.unique((tu.repeat_interleave(10) << 32) | te_items.reshape(-1))
    ip_te, it_te = ops.build_user_csr((te_key >> 32).to(torch.int32), (te_key & 0xFFFFFFFF).to(torch.int32), U)
    t = timeit(lambda: ops.build_candidates(ip_te, it_te, indptr, csr, tu, I, 1000, 3), iters=3, warm=1)
    print(f"build_candidates 100k x 1000       {t:8.3f} ms   {1e8 / t / 1e6:8.2f} G candidates/s")


if __name__ == "__main__":
    print(torch.cuda.get_device_name(0), torch.version.hip)
    t0 = time.time()
    which = sys.argv[1:] or ["phases", "plan"]
    if "mem" in which:
        membench()
    if "phases" in which:
        phases()
        phases(B=1 << 16)
        phases(I=1_000_000)
    if "small" in which:
        phases(U=100_000)                   # P table cache resident: memory- or structure-bound?
    if "plan" in which:
        plan_cost()
    if "frontend" in which:
        frontend()
    print("probe wall", time.time() - t0)
