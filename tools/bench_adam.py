#!/usr/bin/env python
"""MF + BPR with torch.optim.Adam semantics on one GPU: the phase kernels + the dense optimiser pass (every row of both
tables in every step), the phase kernels + the exact lazy form (ops.LazyAdam), and the STAGED step whose row owners apply
Adam themselves (round 3).  BASELINE configs[1] shapes and configs[2] table shapes; the fraction is against SURVEY 8d's
Adam byte model (72 d + 12 = 4620 B per interaction at d = 64):   python tools/bench_adam.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from daisyrec_amd import ops  # noqa: E402
from daisyrec_amd.model.AbstractRecommender import _AdamState  # noqa: E402

dev = torch.device("cuda")
ADAM_BYTES = 72 * 64 + 12
SHAPES = (("configs[1] shapes (1M x 100K)", 1_000_000, 100_000, 50_000_000, 1 << 21),
          ("configs[2] shapes (10M x 1M)", 10_000_000, 1_000_000, 50_000_000, 1 << 21))
MODES = tuple(os.environ.get("ADAM_MODES", "dense,lazy,staged").split(","))          # e.g. ADAM_MODES=staged
for name, U, I, nnz, B in SHAPES[int(os.environ.get("ADAM_FIRST", "0")):int(os.environ.get("ADAM_SHAPES", "2"))]:
    d = 64
    triples = bench.synth_triples(U, I, nnz, 2022, dev)
    n = triples.shape[0]
    for what in MODES:
        if what == "staged":
            index = ops.TrainIndex(triples, U, I, user_sorted=True)
            plan = ops.EpochPlan(n, U, I, device=dev).build_indexed(index, B, order="feistel", seed=1, epoch=0)
            mode = ops.ITEM_MODES["fused"]
        else:
            index = None
            plan = ops.EpochPlan(n, U, I, device=dev).build(triples, B, order="feistel", seed=1, epoch=0, user_sorted=True)
            mode = ops.ITEM_MODES["chunked"]
        nb = min(plan.num_batches, 12)
        g = torch.Generator(device=dev)
        g.manual_seed(1)
        Q = torch.empty(I, d, device=dev).normal_(0.0, 0.01, generator=g)
        P = torch.empty(U, d, device=dev).normal_(0.0, 0.01, generator=g)
        ctx = ops.BprContext(B, d, U, I, device=dev)
        st = _AdamState(P, Q, 0.001, None, kind="adam", max_steps=64, lazy=(what == "lazy"))

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
        print(f"{name}, B={B}: {what:6s} Adam {ms:.3f} ms/step (flush included)  {B / ms / 1e6:.3f} G interactions/s  "
              f"frac of the 8 TB/s roof by the {ADAM_BYTES} B model {ADAM_BYTES * B / (ms * 1e-3) / 8e12:.3f}", flush=True)
        ctx.close()
        plan.close()
        if index is not None:
            index.close()
        del st, P, Q
    del triples
    torch.cuda.empty_cache()
