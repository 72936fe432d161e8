#!/usr/bin/env python
"""LightGCN training throughput on one MI355X (BASELINE configs[4] shape:
Amazon-Book 52 643 users x 91 599 items, 2.38 M training interactions, d=64, 3 layers).

A step = propagation (L sparse x dense products over the WHOLE graph, recomputed per batch like the
reference) + the MF loss/gradient kernels on the batch + backprop (L more products) + dense Adam.
The products are HBM-bound: per stored entry one 4*d-byte row gather + 20 B of (row key, column,
value); per node one row write.  roofline = those algorithmic bytes / time vs 8 TB/s.

    python tools/bench_lightgcn.py > profiles/rNN_bench_lightgcn.txt
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp
import torch

from daisyrec_amd import ops
from daisyrec_amd.model.LightGCNRecommender import LightGCN

U, I, NNZ, D, L = 52643, 91599, 2380730, 64, 3
dev = torch.device("cuda", 0)


def synth():
    rng = np.random.default_rng(0)
    w = 1.0 / np.arange(1, I + 1) ** 0.8                       # popularity-skewed items
    gi = rng.choice(I, size=NNZ, p=w / w.sum())
    gu = rng.integers(0, U, NNZ)
    return gu.astype(np.int64), gi.astype(np.int64)


def main():
    gu, gi = synth()
    import logging
    cfg = dict(gpu="0", logger=logging.getLogger("b"), epochs=1, lr=0.01, topk=50, user_num=U, item_num=I,
               inter_matrix=sp.coo_matrix((np.ones(NNZ, np.float32), (gu, gi)), shape=(U, I)), factors=D,
               num_layers=L, reg_1=0.0, reg_2=0.0, loss_type="BPR", optimizer="default", init_method="default",
               early_stop=False, progress=False)
    torch.manual_seed(0)
    model = LightGCN(cfg)
    E0 = model._ego()
    t0 = time.perf_counter()
    graph = model._adj()
    torch.cuda.synchronize()
    print(json.dumps({"graph": {"nodes": U + I, "entries": graph.nnz, "bytes": graph.nbytes,
                                "build_ms": (time.perf_counter() - t0) * 1e3}}), flush=True)
    ev = lambda: torch.cuda.Event(enable_timing=True)
    # ---- the sparse x dense product alone
    X = torch.randn(U + I, D, device=dev)
    graph.spmm(X)
    e0, e1 = ev(), ev()
    e0.record()
    for _ in range(20):
        graph.spmm(X)
    e1.record()
    torch.cuda.synchronize()
    spmm_ms = ms = e0.elapsed_time(e1) / 20
    alg = graph.nnz * (4 * D + 20) + (U + I) * 4 * D * 2          # gathers + entry metadata + memset + row writes
    print(json.dumps({"spmm": {"ms": ms, "algorithmic_GB": alg / 1e9, "GBps": alg / ms / 1e6,
                               "frac_of_hbm_peak": alg / ms / 1e6 / 8000.0,
                               "G_rows_per_s": graph.nnz / ms / 1e6}}), flush=True)
    # ---- training steps
    loss_id = ops.loss_id("BPR")
    out, G, dE0 = torch.empty_like(E0), torch.empty_like(E0), torch.zeros_like(E0)
    m, v = torch.zeros_like(model._flat), torch.zeros_like(model._flat)
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    for B, steps in ((256, 50), (4096, 50), (65536, 30), (1 << 20, 10)):
        u = torch.randint(0, U, (B,), device=dev, generator=g, dtype=torch.int32)
        i = torch.randint(0, I, (B,), device=dev, generator=g, dtype=torch.int32)
        j = torch.randint(0, I, (B,), device=dev, generator=g, dtype=torch.int32)
        ctx = ops.BprContext(B, D, U, I)
        def step(t):
            model._batch_grads(ctx, E0, out, G, dE0, u, i, j, loss_id)
            ops.adam_dense(model._flat, dE0.view(-1), m, v, 0.01, t)
        for t in range(1, 4):
            step(t)
        torch.cuda.synchronize()
        e0, e1 = ev(), ev()
        e0.record()
        for t in range(4, 4 + steps):
            step(t)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / steps
        print(json.dumps({"B": B, "ms_per_step": ms, "samples_per_s": B / ms * 1e3,
                          "propagation_share": 2 * L * spmm_ms / ms}), flush=True)
        ctx.close()
    if "--no-cpu" not in sys.argv:
        from oracle.torch_port import TorchLightGCN
        from oracle import lightgcn_numpy as LG
        indptr, col, val = LG.norm_adj_csr(gu, gi, U, I)
        rows = np.repeat(np.arange(U + I), np.diff(indptr))
        adj = torch.sparse_coo_tensor(torch.as_tensor(np.stack([rows, col.astype(np.int64)])), torch.as_tensor(val),
                                      (U + I, U + I)).coalesce()
        torch.manual_seed(0)
        mdl = TorchLightGCN(U, I, D, L, adj)
        gg = torch.Generator()
        gg.manual_seed(1)
        B, steps = 4096, 3
        bt = [(torch.randint(0, U, (B,), generator=gg), torch.randint(0, I, (B,), generator=gg),
               torch.randint(0, I, (B,), generator=gg)) for _ in range(steps + 1)]
        mdl.step(*bt[0])
        t0 = time.perf_counter()
        for b in bt[1:]:
            mdl.step(*b)
        dt = time.perf_counter() - t0
        print(json.dumps({"cpu_baseline": {"ms_per_step": dt / steps * 1e3, "samples_per_s": steps * B / dt,
                                           "cores": torch.get_num_threads(), "kind": "port",
                                           "sample": f"{steps} Adam steps at B={B} (oracle/torch_port.py: TorchLightGCN, "
                                                     f"torch.sparse.mm propagation per batch), {dt:.1f}s"}}), flush=True)


if __name__ == "__main__":
    main()
