#!/usr/bin/env python
"""MF + Adam at the batch sizes real daisyRec runs use: the epoch as ONE enqueue (daisy_bpr_fit_epoch_adam, round 5) against
the round-4 loop that drove every step from Python (two ctypes calls per batch) - us per step, ml-100k shapes and BASELINE
configs[1] tables:   python tools/adam_small.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from daisyrec_amd import ops  # noqa: E402

dev = torch.device("cuda")
for name, U, I, nnz, d in (("ml-100k shapes", 943, 1152, 78_363, 32), ("configs[1] tables", 1_000_000, 100_000, 4_000_000, 64)):
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    u = torch.sort(torch.randint(0, U, (nnz,), device=dev, generator=g)).values
    tri = torch.stack([u, torch.randint(0, I, (nnz,), device=dev, generator=g),
                       torch.randint(0, I, (nnz,), device=dev, generator=g)], 1).to(torch.int32).contiguous()
    for B in (256, 1024, 4096):
        n = min(nnz, B * 600) // B * B
        tr = tri[:n].contiguous()
        index = ops.TrainIndex(tr, U, I, user_sorted=True)
        plan = ops.EpochPlan(n, U, I, device=dev).build_indexed(index, B, order="feistel", seed=1, epoch=0)
        nb = plan.num_batches
        res = {}
        if B <= ops.SMALL_BATCH_MAX:
            # round 6: the sorted plan -> every step of the epoch inside one persistent workgroup (csrc/bpr_small.hip), SGD and Adam
            splan = ops.EpochPlan(n, U, I, device=dev).build(tr, B, order="feistel", seed=1, epoch=0, user_sorted=True)
            for opt in ("sgd", "adam"):
                P = torch.empty(U, d, device=dev).normal_(0.0, 0.01, generator=g)
                Q = torch.empty(I, d, device=dev).normal_(0.0, 0.01, generator=g)
                ctx = ops.BprContext(B, d, U, I, device=dev)
                adam = ops.LazyAdam(P, Q, 0.001, 4 * nb) if opt == "adam" else None
                assert ops.LazyAdam.small_epoch_supported(ctx, splan)

                def epoch():
                    if adam is not None:
                        adam.fit_epoch(ctx, splan, 1e-3, 1e-3)
                    else:
                        ctx.fit_epoch_sgd(splan, P, Q, 0.01, 1e-3, 1e-3)

                epoch()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                epoch()
                torch.cuda.synchronize()
                res[f"persistent workgroup, {opt}"] = (time.perf_counter() - t0) / nb * 1e6
                ctx.close()
            splan.close()
        for how in ("python loop", "one enqueue"):
            P = torch.empty(U, d, device=dev).normal_(0.0, 0.01, generator=g)
            Q = torch.empty(I, d, device=dev).normal_(0.0, 0.01, generator=g)
            ctx = ops.BprContext(B, d, U, I, device=dev)
            adam = ops.LazyAdam(P, Q, 0.001, 4 * nb)

            def epoch():
                if how == "one enqueue":
                    adam.fit_epoch(ctx, plan, 1e-3, 1e-3)
                else:
                    for k in range(nb):
                        ctx.set_batch_from_plan(plan, k)
                        adam.staged_step(ctx, 1e-3, 1e-3)
                    adam.flush()

            epoch()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            epoch()
            torch.cuda.synchronize()
            res[how] = (time.perf_counter() - t0) / nb * 1e6
            ctx.close()
        print(f"{name}, d={d}, B={B:5d}, {nb} steps per epoch: " + ", ".join(f"{k} {v:7.1f} us/step" for k, v in res.items()), flush=True)
        plan.close(); index.close()
