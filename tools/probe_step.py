#!/usr/bin/env python
"""Kernel probe: one workload (c2 = BASELINE configs[1]; c3s = configs[2] table shapes with the interaction count cut
to 100 M; c3r = one rank's share of configs[2]), the staged step through the Python step loop of bench.py, HIP-event
timings of the plan build and of the steps alone.  Run it under `rocprofv3 --kernel-trace --stats` for the per-kernel
split (tools/probe_run.sh); knobs of the library are environment variables or -D flags of a development
build selected with DAISY_LIB_OVERRIDE (tools/devlib.sh)."""
import os
import sys

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


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 21
    tag = os.environ.get("TAG", "")
    U, I, nnz = {"c2": (1_000_000, 100_000, 50_000_000), "c3s": (10_000_000, 1_000_000, 100_000_000),
                 "c3r": (1_250_000, 1_000_000, 62_500_000)}[wl]
    dev, d = torch.device("cuda"), 64
    triples = bench.synth_triples(U, I, nnz, 2022, dev)
    n = triples.shape[0]
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    Q = torch.empty(I, d, device=dev).normal_(0.0, 0.01, generator=g)
    P = torch.empty(U, d, device=dev).normal_(0.0, 0.01, generator=g)
    ctx = ops.BprContext(B, d, U, I, device=dev)
    if os.environ.get("PROBE_PSTREAM"):          # 0 / 1: force the nontemporal user rows off / on (default: by table size)
        ctx.set_p_stream(bool(int(os.environ["PROBE_PSTREAM"])))
    index = ops.TrainIndex(triples, U, I, user_sorted=True)
    plan = ops.EpochPlan(n, U, I, device=dev)
    ep = [0]

    def build():
        ep[0] += 1
        plan.build_indexed(index, B, order="feistel", seed=1, epoch=ep[0])

    build()
    plan_ms = ev_time(build, 3)
    nb = n // B
    k = [0]

    phases = os.environ.get("PROBE_PHASES", "")     # "grad": the multi-GPU form (item pass writes gQ + counts, no Q access)
    if phases:
        gQ = torch.zeros_like(Q)
        cnt = torch.zeros(I, 2, device=dev)

    def step():
        ctx.set_batch_from_plan(plan, k[0] % nb)
        if phases:
            ctx.staged_prenorm(P)
            ctx.staged_user(P, Q, 0.01, 1e-3, 1e-3)
            ctx.staged_item(0.01, 1e-3, 1e-3, gQ=gQ, cnt=cnt)
        else:
            ctx.sgd_step(P, Q, 0.01, 1e-3, 1e-3, item_mode=ops.ITEM_MODES["fused"])
        k[0] += 1

    for _ in range(3):
        step()
    ms = ev_time(step, steps)
    ov = os.environ.get("PROBE_OVERLAP", "")
    if ov:
        # epochs with the NEXT epoch's plan built on a side stream while this epoch's steps run
        lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
        main = torch.cuda.Stream(device=dev, priority=-1) if ov == "prio" else torch.cuda.current_stream()
        side = torch.cuda.Stream(device=dev, priority=0)
        plans = [plan, ops.EpochPlan(n, U, I, device=dev)]
        plans[1].build_indexed(index, B, order="feistel", seed=1, epoch=99)
        torch.cuda.synchronize()

        def epochs(E):
            cur = 0
            ready = None
            with torch.cuda.stream(main):
                for e in range(E):
                    if ready is not None:
                        main.wait_event(ready)
                    side.wait_stream(main)                   # the other plan's last epoch has been consumed
                    with torch.cuda.stream(side):
                        plans[cur ^ 1].build_indexed(index, B, order="feistel", seed=1, epoch=100 + e)
                        ready = side.record_event()
                    for kk in range(nb):
                        ctx.set_batch_from_plan(plans[cur], kk)
                        ctx.sgd_step(P, Q, 0.01, 1e-3, 1e-3, item_mode=ops.ITEM_MODES["fused"])
                    cur ^= 1

        epochs(1)
        torch.cuda.synchronize()
        E = 3
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(main)
        epochs(E)
        b.record(main)
        torch.cuda.synchronize()
        tot_ov = a.elapsed_time(b) / (E * nb)
        print(f"[{wl} {tag}] overlap={ov}: {tot_ov * 1e3:.1f} us/step incl. plan  frac(total) {1548 * B / (tot_ov * 1e-3) / 8e12:.3f}  "
              f"(sequential: {(ms + plan_ms / nb) * 1e3:.1f})", flush=True)
    loss, bad = (float(x) for x in ctx.epoch_acc.cpu())
    per_step_plan = plan_ms / nb
    tot = ms + per_step_plan
    print(f"[{wl}{' ' + tag if tag else ''}] B={B} nb={nb} plan {plan_ms:.3f} ms/epoch ({per_step_plan * 1e3:.1f} us/step)  "
          f"step {ms * 1e3:.1f} us  total {tot * 1e3:.1f} us/step  frac(steps) {1548 * B / (ms * 1e-3) / 8e12:.3f}  "
          f"frac(total) {1548 * B / (tot * 1e-3) / 8e12:.3f}  loss_sum {loss:.6e} bad {bad}", flush=True)


if __name__ == "__main__":
    main()
