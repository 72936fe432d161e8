#!/usr/bin/env python
"""Benchmark of the MF + BPR training hot path on MI355X (contract in the task brief).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one batch-synchronous SGD step (zero_grad / calc_loss / backward / step of the
reference, AbstractRecommender.py:119-126) over one batch of B interactions per GPU, B stated
in `config`.  Inputs (tables, triples, epoch permutation state) are resident in HBM when the
timed region starts; the per-epoch device shuffle that falls inside the timed steps IS timed.

N=1  : BASELINE.json configs[1]  U=1M, I=100k, nnz=50M, d=64 (uniform ids), SGD, lr .01, reg .001.
N>1  : weak scaling of that shard — every rank owns 1M users / 50M interactions (users sharded,
       Q replicated, RCCL all-reduce of the item gradient).  `--workload c3` instead splits
       BASELINE configs[2] (10M x 1M x 500M) over the ranks (total work fixed: strong scaling).

Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_PER_INTERACTION_SGD = lambda d: 24 * d + 12      # 3 row reads + 3 row writes + 3 int32
HBM_PEAK_GBS = 8000.0                                        # MI355X_MICROARCH.md: 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="default: one epoch of full batches")
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--batch", type=int, default=1 << 21, help="interactions per GPU per step")
    ap.add_argument("--workload", default="auto", choices=["auto", "c2", "c3", "tiny"])
    ap.add_argument("--item-mode", default="fused", choices=["fused", "chunked", "atomic", "sorted"])
    ap.add_argument("--plan", default="auto", choices=["auto", "indexed", "sorted"],
                    help="epoch plan layout: indexed = partitioned (staged step only), sorted = radix-sorted")
    ap.add_argument("--dist", default="uniform", choices=["uniform", "zipf"])
    ap.add_argument("--reg", type=float, default=0.001)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--overlap-plan", type=int, default=0,
                    help="build the next epoch's plan on a side stream (two plans in ping-pong; measured slower with the partitioned plan)")
    ap.add_argument("--cpu-steps", type=int, default=3)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend (nccl = RCCL; gloo only to exercise the N>1 code path on one GPU)")
    return ap.parse_args()


def synth_triples(U, I, nnz, seed, device, dist_kind="uniform"):
    """User-sorted, duplicate-free (u,i) pairs + one uniform negative per interaction drawn by
    the HIP sampler from the complement of the user's row."""
    from daisyrec_amd import ops
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    u = torch.randint(0, U, (nnz,), device=device, generator=g, dtype=torch.int64)
    if dist_kind == "uniform":
        i = torch.randint(0, I, (nnz,), device=device, generator=g, dtype=torch.int64)
    else:  # Zipf(1.0) item popularity truncated at I
        w = 1.0 / torch.arange(1, I + 1, device=device, dtype=torch.float64)
        cdf = torch.cumsum(w / w.sum(), 0)
        i = torch.searchsorted(cdf, torch.rand(nnz, device=device, generator=g, dtype=torch.float64))
        i = i.clamp_(max=I - 1)
    key = torch.unique((u << 32) | i)                     # sorted by (user, item), duplicates dropped
    del u, i
    users = (key >> 32).to(torch.int32)
    items = (key & 0xFFFFFFFF).to(torch.int32)
    del key
    indptr, csr = ops.build_user_csr(users, items, U)
    triples = torch.stack([users, items, torch.zeros_like(items)], 1).contiguous()
    del users, items
    ops.resample_neg_per_interaction(indptr, csr, I, triples, seed, 0)
    torch.cuda.synchronize()
    return triples


def cpu_baseline(U, I, d, B, steps, reg):
    """The reference's CPU/PyTorch path (oracle/torch_port.py restates it with the same stock
    ops) timed on this host's cores on a bounded sample: `steps` steps at the SAME batch size."""
    from oracle.torch_port import TorchMFBPR
    torch.manual_seed(2022)
    m = TorchMFBPR(U, I, d, 0.01, reg, reg)
    g = torch.Generator()
    g.manual_seed(1)
    batches = [(torch.randint(0, U, (B,), generator=g), torch.randint(0, I, (B,), generator=g),
                torch.randint(0, I, (B,), generator=g)) for _ in range(steps + 1)]
    m.step(*batches[0])                                    # warm-up (allocations, thread pool)
    t0 = time.perf_counter()
    for b in batches[1:]:
        m.step(*b)
    dt = time.perf_counter() - t0
    return {"value": steps * B / dt, "unit": "interactions/s", "cores": torch.get_num_threads(),
            "kind": "port",
            "sample": f"{steps} SGD steps at B={B} on U={U}, I={I}, d={d} (oracle/torch_port.py: "
                      f"nn.Embedding + autograd + optim.SGD, dense grads like the reference), {dt:.1f}s"}


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    assert torch.cuda.is_available(), "bench.py needs a HIP device (no CPU fallback)"
    local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)   # RCCL
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    from daisyrec_amd import ops
    from daisyrec_amd.sharding import UserShardedBprTrainer

    d = 64
    wl = a.workload if a.workload != "auto" else "c2"
    if wl == "c2":
        U_loc, I, nnz_loc, scaling = 1_000_000, 100_000, 50_000_000, "weak"
        name = ("BASELINE configs[1]: MF+BPR synthetic 1M users x 100K items x 50M nnz, d=64" if world == 1 else
                f"weak scaling of BASELINE configs[1]: {world}M users x 100K items x {50 * world}M nnz, d=64, "
                f"user-sharded over {world} GPUs (1M users / 50M nnz per GPU)")
    elif wl == "c3":
        U_tot, I, nnz_tot, scaling = 10_000_000, 1_000_000, 500_000_000, "strong"
        U_loc, nnz_loc = U_tot // world, nnz_tot // world
        name = f"BASELINE configs[2]: 10M users x 1M items x 500M nnz, d=64, user-sharded over {world} GPU(s)"
    else:
        U_loc, I, nnz_loc, scaling = 20_000, 5_000, 1_000_000, "weak"
        name = "tiny smoke workload (NOT a BASELINE config)"
    B = min(a.batch, nnz_loc)
    lr, reg = 0.01, a.reg

    # ---- data + model resident in HBM -------------------------------------------------
    triples = synth_triples(U_loc, I, nnz_loc, 2022 + rank, dev, a.dist)      # LOCAL user ids
    n = triples.shape[0]
    g = torch.Generator(device=dev)
    g.manual_seed(2022)
    Q = torch.empty(I, d, device=dev).normal_(0.0, 0.01, generator=g)         # identical on every rank
    g.manual_seed(7 + rank)
    P = torch.empty(U_loc, d, device=dev).normal_(0.0, 0.01, generator=g)
    ctx = ops.BprContext(B, d, U_loc, I, device=dev)
    plan = ops.EpochPlan(n, U_loc, I, device=dev)
    item_mode = ops.ITEM_MODES[a.item_mode]
    trainer = UserShardedBprTrainer(ctx, P, Q, 0, lr, reg, reg, item_mode=item_mode) if world > 1 else None
    user_sorted = ops.triples_user_sorted(triples)      # synthetic triples are generated in CSR order
    plan_kind = a.plan if a.plan != "auto" else ("indexed" if a.item_mode == "fused" else "sorted")
    index = ops.TrainIndex(triples, U_loc, I, user_sorted=user_sorted) if plan_kind == "indexed" else None
    full_batches = n // B                      # the bench steps over full batches only (fixed B per step)
    if world > 1:                              # shards differ by a few interactions (dedup): agree on the
        fb = torch.tensor([full_batches], device=dev, dtype=torch.int64)     # count, every rank must take the same steps
        dist.all_reduce(fb, op=dist.ReduceOp.MIN)
        full_batches = int(fb.cpu())
    if a.steps is None:
        a.steps = full_batches                 # one epoch: exactly one plan build inside the timed region

    # Two plans in ping-pong: while epoch e trains on the main stream, the plan of epoch e+1
    # (DataLoader(shuffle=True) on the device: a fresh keyed permutation, radix sorts lay the
    # epoch out batch by batch grouped by user / by item) is built on a side stream.
    plans = [plan, ops.EpochPlan(n, U_loc, I, device=dev)] if a.overlap_plan else [plan]
    side = torch.cuda.Stream(device=dev) if a.overlap_plan else None
    state = {"epoch": 0, "k": None, "cur": 0, "ready": None}

    def build_plan(slot, epoch):
        if index is not None:
            plans[slot].build_indexed(index, B, order="feistel", seed=2022 + rank, epoch=epoch)
        else:
            plans[slot].build(triples, B, order="feistel", seed=2022 + rank, epoch=epoch, user_sorted=user_sorted)

    def build(slot, epoch, stream=None):
        if stream is None:
            build_plan(slot, epoch)
            return None
        stream.wait_stream(torch.cuda.current_stream())      # the slot's previous epoch has been consumed
        with torch.cuda.stream(stream):
            build_plan(slot, epoch)
            return stream.record_event()

    def step():
        if state["k"] is None or state["k"] >= full_batches:
            if not a.overlap_plan:
                build(0, state["epoch"])
            elif state["k"] is None:                              # cold start
                build(0, state["epoch"])
                state["cur"] = 0
                state["ready"] = build(1, state["epoch"] + 1, side)
            else:                                                 # steady state: swap, prefetch the next
                torch.cuda.current_stream().wait_event(state["ready"])
                state["cur"] ^= 1
                state["ready"] = build(state["cur"] ^ 1, state["epoch"] + 1, side)
            state["epoch"] += 1
            state["k"] = 0
        k = state["k"]
        state["k"] += 1
        pl = plans[state["cur"]]
        if trainer is None:
            ctx.set_batch_from_plan(pl, k)
            ctx.sgd_step(P, Q, lr, reg, reg, item_mode=item_mode)
        else:
            trainer.step_from_plan(pl, k)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    # start the timed region on an epoch boundary: every timed epoch then contains exactly one
    # plan build (with --overlap-plan: the build of the NEXT epoch, running beside the steps)
    state["k"] = full_batches if (a.overlap_plan and state["k"] is not None) else None
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]
    barrier()
    t0 = time.perf_counter()
    for k in range(a.steps):
        ev[k][0].record()
        step()
        ev[k][1].record()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.cpu())
    loss_sum, nan_cnt = (float(x) for x in ctx.epoch_acc.cpu())
    assert nan_cnt == 0 and loss_sum == loss_sum, "NaN loss during the benchmark"
    step_ms = sorted(s.elapsed_time(e) for s, e in ev)
    gpu_ms_mean = sum(step_ms) / len(step_ms)

    if rank == 0:
        value = a.steps * B * world / dt
        algo = ALGO_BYTES_PER_INTERACTION_SGD(d) * B                       # bytes per step per GPU
        eff_ms = max(gpu_ms_mean, dt / a.steps * 1e3) if world == 1 else gpu_ms_mean   # never better than wall
        achieved = algo / (eff_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            try:
                t = json.load(open(tpath))          # per-interaction figure of the profiled run x this run's batch
                traffic = t["hbm_bytes_per_interaction"] * B if "hbm_bytes_per_interaction" in t else None
            except Exception:
                traffic = None
        out = {
            "metric": "BPR training interactions/sec at d=64; achieved HBM GB/s vs peak",
            "value": value, "unit": "interactions/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": name, "batch_per_gpu": B, "global_batch": B * world, "d": d,
                       "optimizer": "sgd", "lr": lr, "reg_1": reg, "reg_2": reg, "loss": "BPR",
                       "item_mode": a.item_mode, "id_distribution": a.dist, "interactions_per_gpu": n,
                       "parallelism": f"user-sharded dp{world}" if world > 1 else "single GPU",
                       "plan_bytes": plan.nbytes * len(plans), "plan_overlapped": bool(a.overlap_plan),
                       "plan_layout": plan_kind, "index_bytes": index.nbytes if index is not None else 0,
                       "semantics": "batch-synchronous (autograd + SGD.step equivalent), shuffle=True (device Feistel permutation per epoch)"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": ("one SGD step = k_unorm + k_user_fused + k_reduce_partials + k_item_grad_chunked + k_user_commit + k_item_apply"
                                    if a.item_mode == "fused" else
                                    "one SGD step = k_fwd + k_reduce_partials + k_item_grad_" + a.item_mode + " + k_user + k_item_apply")
                                   + " (+ the epoch plan build amortised over its batches)",
                         "algorithmic_bytes_per_interaction": ALGO_BYTES_PER_INTERACTION_SGD(d),
                         "gpu_ms_per_step_events": gpu_ms_mean, "gpu_ms_per_step_median": step_ms[len(step_ms) // 2]},
        }
        if world == 1 and not a.no_cpu_baseline:
            del triples
            out["cpu_baseline"] = cpu_baseline(U_loc, I, d, B, a.cpu_steps, reg)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
