#!/usr/bin/env python
"""MF + BPR with torch.optim.Adam semantics: the dense optimiser pass (every row of both tables in every step) against
the exact lazy form (ops.LazyAdam), C2 shapes and BASELINE configs[2] shapes on one GPU: python tools/bench_adam.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from daisyrec_amd import ops  # noqa: E402
from daisyrec_amd.model.AbstractRecommender import _AdamState  # noqa: E402

dev = torch.device("cuda")
for name, U, I, nnz, B in (("C2 shapes (1M x 100K)", 1_000_000, 100_000, 20_000_000, 1 << 20),
                           ("configs[2] shapes (10M x 1M)", 10_000_000, 1_000_000, 40_000_000, 1 << 21)):
    d = 64
    triples = bench.synth_triples(U, I, nnz, 2022, dev)
    n = triples.shape[0]
    plan = ops.EpochPlan(n, U, I, device=dev).build(triples, B, order="feistel", seed=1, epoch=0, user_sorted=True)
    nb = min(plan.num_batches, 12)
    for lazy in (False, True):
        g = torch.Generator(device=dev)
        g.manual_seed(1)
        Q = torch.empty(I, d, device=dev).normal_(0.0, 0.01, generator=g)
        P = torch.empty(U, d, device=dev).normal_(0.0, 0.01, generator=g)
        ctx = ops.BprContext(B, d, U, I, device=dev)
        st = _AdamState(P, Q, 0.001, None, kind="adam", max_steps=64, lazy=lazy)
        mode = ops.ITEM_MODES["chunked"]

        def epoch():
            for k in range(nb):
                ctx.set_batch_from_plan(plan, k)
                st.step(ctx, P, Q, 1e-3, 1e-3, 0, mode)
            st.flush()

        epoch()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        epoch()
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / nb
        print(f"{name}, B={B}: {'lazy ' if lazy else 'dense'} Adam {ms:.3f} ms/step (flush included)  "
              f"{B / ms / 1e6:.3f} G interactions/s", flush=True)
        ctx.close()
        del st, P, Q
    plan.close()
    del triples
    torch.cuda.empty_cache()
