#!/usr/bin/env python
"""SURVEY.md §8(d) asks for the headline metric under several batch sizes, both id distributions,
reg=0 and Adam.  One GPU, BASELINE configs[1] (U=1M, I=100k, nnz=50M, d=64); each line is
interactions/s over `steps` consecutive batches of a shuffled epoch (plan build reported apart).

    python tools/bench_matrix.py > profiles/rNN_bench_matrix.txt
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from daisyrec_amd import ops
from daisyrec_amd.model.AbstractRecommender import _AdamState

U, I, NNZ, D = 1_000_000, 100_000, 50_000_000, 64
dev = torch.device("cuda", 0)


def run(triples, B, steps, opt="sgd", reg=1e-3, item_mode="chunked", label=""):
    n = min(steps * B, triples.shape[0])
    if n < triples.shape[0]:                      # a prefix of a shuffled epoch = a random sample
        g = torch.Generator(device=dev)
        g.manual_seed(7)
        sel = torch.randperm(triples.shape[0], device=dev, generator=g)[:n]
        tri = triples[sel].contiguous()
    else:
        tri = triples
    torch.manual_seed(2022)
    P = torch.empty(U, D, device=dev).normal_(0, 0.01)
    Q = torch.empty(I, D, device=dev).normal_(0, 0.01)
    ctx = ops.BprContext(B, D, U, I, device=dev)
    plan = ops.EpochPlan(n, U, I, device=dev)
    us = ops.triples_user_sorted(tri)
    mode = ops.ITEM_MODES[item_mode]
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    res = []
    for rep in range(2):                           # rep 0 = warm-up
        torch.cuda.synchronize()
        e0.record()
        plan.build(tri, B, order="feistel", seed=2022, epoch=rep, n_triples=n, user_sorted=us)
        e1.record()
        if opt == "sgd":
            ctx.fit_epoch_sgd(plan, P, Q, 0.01, reg, reg, item_mode=mode)
            nb = plan.num_batches
        else:
            adam = _AdamState(P, Q, 0.001)
            nb = min(plan.num_batches, steps)
            for k in range(nb):
                ctx.set_batch_from_plan(plan, k)
                adam.step(ctx, P, Q, reg, reg, 0, mode)
        e2.record()
        torch.cuda.synchronize()
        res = [e0.elapsed_time(e1), e1.elapsed_time(e2), nb]
    plan_ms, fit_ms, nb = res
    inter = min(nb * B, n)
    out = {"case": label, "B": B, "steps": nb, "optimizer": opt, "reg": reg, "item_mode": item_mode,
           "ms_per_step": fit_ms / nb, "plan_ms": plan_ms,
           "G_inter_per_s_steps_only": inter / fit_ms / 1e6,
           "G_inter_per_s_with_plan": inter / (fit_ms + plan_ms) / 1e6,
           "roofline_frac_with_plan": inter / (fit_ms + plan_ms) / 1e6 * 1548 / 8000}
    print(json.dumps(out), flush=True)
    ctx.close()
    plan.close()


def main():
    t0 = time.time()
    tri = bench.synth_triples(U, I, NNZ, 2022, dev, "uniform")
    run(tri, 1 << 20, 47, label="uniform B=1M sgd")
    run(tri, 1 << 20, 47, reg=0.0, label="uniform B=1M sgd reg=0")
    run(tri, 1 << 16, 400, label="uniform B=65536 sgd")
    run(tri, 256, 4000, label="uniform B=256 sgd (reference default batch)")
    run(tri, 256, 4000, item_mode="sorted", label="uniform B=256 sgd, reproducible mode")
    run(tri, 1 << 20, 8, opt="adam", label="uniform B=1M dense adam")
    del tri
    tri = bench.synth_triples(U, I, NNZ, 2022, dev, "zipf")
    run(tri, 1 << 20, 47, label="zipf(1.0) items B=1M sgd")
    run(tri, 1 << 16, 400, label="zipf(1.0) items B=65536 sgd")
    print(f"# total {time.time() - t0:.1f}s", flush=True)


if __name__ == "__main__":
    main()
