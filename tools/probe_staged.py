#!/usr/bin/env python
"""A/B timings of the staged step and the partitioned plan (HIP events), C2 / C3 shapes on one GPU."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from daisyrec_amd import ops  # noqa: E402


def ev_time(fn, reps):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def run(U, I, nnz, B, modes, tag):
    dev = torch.device("cuda")
    d = 64
    triples = bench.synth_triples(U, I, nnz, 2022, dev)
    n = triples.shape[0]
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    Q = torch.empty(I, d, device=dev).normal_(0.0, 0.01, generator=g)
    P = torch.empty(U, d, device=dev).normal_(0.0, 0.01, generator=g)
    ctx = ops.BprContext(B, d, U, I, device=dev)
    nb = n // B
    t0 = time.perf_counter()
    index = ops.TrainIndex(triples, U, I, user_sorted=True)
    torch.cuda.synchronize()
    print(f"[{tag}] n={n} B={B} nb={nb}  index build {1e3 * (time.perf_counter() - t0):.1f} ms, {index.nbytes / n:.1f} B/interaction")
    plan_i, plan_s = ops.EpochPlan(n, U, I, device=dev), ops.EpochPlan(n, U, I, device=dev)
    ep = [0]

    def bi():
        ep[0] += 1
        plan_i.build_indexed(index, B, order="feistel", seed=1, epoch=ep[0])

    def bs():
        ep[0] += 1
        plan_s.build(triples, B, order="feistel", seed=1, epoch=ep[0], user_sorted=True)

    bi(); bs()
    print(f"[{tag}] plan build: indexed {ev_time(bi, 3):.3f} ms ({plan_i.nbytes / n:.1f} B/int)   sorted {ev_time(bs, 3):.3f} ms ({plan_s.nbytes / n:.1f} B/int)")
    steps = min(nb, 40)
    for name, plan, mode in modes:
        pl = plan_i if plan == "i" else plan_s
        k = [0]

        def step():
            ctx.set_batch_from_plan(pl, k[0] % steps)
            ctx.sgd_step(P, Q, 0.01, 1e-3, 1e-3, item_mode=ops.ITEM_MODES[mode])
            k[0] += 1

        for _ in range(3):
            step()
        ms = ev_time(step, steps)
        print(f"[{tag}] {name:28s} {ms:.4f} ms/step  {B / ms / 1e6:.3f} G/s  frac {1548 * B / (ms * 1e-3) / 8e12:.3f}", flush=True)
    ctx.close(); plan_i.close(); plan_s.close(); index.close()


def rank_share(B):
    """Per-rank compute of the 8-GPU staged protocol at BASELINE configs[2] (U/8 users, I=1M, nnz/8), collectives
    replaced by nothing: what is left to overlap or expose the exchange against."""
    from daisyrec_amd.sharding import UserShardedBprTrainer
    dev = torch.device("cuda")
    U, I, nnz, d = 1_250_000, 1_000_000, 62_500_000, 64
    triples = bench.synth_triples(U, I, nnz, 2022, dev)
    n = triples.shape[0]
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    Q = torch.empty(I, d, device=dev).normal_(0.0, 0.01, generator=g)
    P = torch.empty(U, d, device=dev).normal_(0.0, 0.01, generator=g)
    B = min(B, n)
    ctx = ops.BprContext(B, d, U, I, device=dev)
    index = ops.TrainIndex(triples, U, I, user_sorted=True)
    plan = ops.EpochPlan(n, U, I, device=dev).build_indexed(index, B, order="feistel", seed=1, epoch=0)
    slices = int(os.environ.get("SLICES", "1"))
    tr = UserShardedBprTrainer(ctx, P, Q, 0, 0.01, 1e-3, 1e-3, slices=slices)       # world 1, no process group: collectives are no-ops
    nb = n // B
    k = [0]

    def step():
        tr.step_from_plan(plan, k[0] % nb)
        k[0] += 1

    for _ in range(2):
        step()
    ms = ev_time(step, 2 * nb)
    wire = 2 * 7 / 8 * I * (d + 2) * 4
    print(f"[c3rank] slices={slices} n={n} B={B}: {ms:.3f} ms/step per rank without collectives ({B / ms / 1e6:.3f} G/s per rank); "
          f"wire per step and rank 2 x {wire / 2 / 1e6:.0f} MB; at 310 GB/s bus bandwidth {wire / 310e9 * 1e3:.2f} ms")


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "c2"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 21
    modes = [("staged/indexed", "i", "fused"), ("staged/sorted-plan", "s", "fused"), ("chunked (r01)", "s", "chunked")]
    if which == "c3rank":      # one rank's share of BASELINE configs[2] on 8 GPUs, through the sharded trainer's phases
        rank_share(B if len(sys.argv) > 2 else 1 << 24)
    elif which == "c2":
        run(1_000_000, 100_000, 50_000_000, B, modes, "c2")
    else:
        run(10_000_000, 1_000_000, 200_000_000, B, modes, "c3-200M")
