#!/usr/bin/env python
"""Benchmark of the MF + BPR training hot path on MI355X (contract in the task brief).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

`python bench.py --gpus N` with N > 1 and no launcher around it spawns its N ranks itself (one process
per GPU, RCCL) and refuses to run when the box has fewer than N GPUs: it never reports a smaller job
under a bigger name.

A "step" is one batch-synchronous SGD step (zero_grad / calc_loss / backward / step of the
reference, AbstractRecommender.py:119-126) over one batch of B interactions per GPU, B stated
in `config`.  Inputs (tables, triples, the per-fit index) are resident in HBM when the timed region
starts; the per-epoch plan builds (the device side of DataLoader(shuffle=True)) that fall inside the
timed steps ARE timed.  Default timed region: two epochs of full batches.

N=1  : BASELINE.json configs[1]  U=1M, I=100k, nnz=50M, d=64 (uniform ids), SGD, lr .01, reg .001 - the line's
       `value`.  The same run then measures, as `secondary`, the pure-HBM regime: the table shapes of configs[2]
       (10M users x 1M items: P 2.56 GB, Q 256 MB, both beyond the 256 MB Infinity Cache) on this one GPU at the same
       2M-interaction steps, with the interaction count cut to 100M (a shorter epoch; what a step moves depends on the
       tables and the batch, not on the number of batches per epoch), and the CPU baseline.
N>1  : BASELINE.json configs[2]  U=10M, I=1M, nnz=500M split by user over the N ranks (strong scaling: total work
       fixed; Q replicated; reduce-scatter / owner apply / all-gather of the item update over RCCL).  The run is
       SELF-DIAGNOSING (nobody has run it on more than one GPU before the driver does): the line carries the RCCL
       version, the ranks' devices, a replica check (a small fit through `MF.fit` sharded over the ranks against the
       same fit on one GPU: epoch losses and Q), the per-step split into compute and exposed exchange, a sweep over
       B_local x exchange slices with the >= 6x verdict per point, and - unless --no-ref - the SAME workload on one GPU
       measured by rank 0 afterwards.  `--workload c2` weak-scales configs[1] instead.

Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import socket
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_PER_INTERACTION_SGD = lambda d: 24 * d + 12      # 3 row reads + 3 row writes + 3 int32
HBM_PEAK_GBS = 8000.0                                        # MI355X_MICROARCH.md: 8.0 TB/s spec
SECONDARY_NNZ = 100_000_000                                  # interactions of the N=1 `secondary` leg (configs[2] tables)


# ------------------------------------------------------------------------------------------------------------------
# The printed line.  The driver keeps an 8 KB tail of stdout: the line carries numbers and short keys only (the prose that
# explains each leg lives in DESIGN.md section 6 under the same names) and prints the `extra` legs BEFORE `secondary` and
# `cpu_baseline`; the verbose record of the same run goes to gpurun_out/bench_full.json (scratch, not tracked).
# ------------------------------------------------------------------------------------------------------------------
_KEEP_STR = {"metric", "unit", "dtype", "data", "scaling", "bound", "kind", "optimizer", "loss", "item_mode", "parallelism",
             "id_distribution", "plan_layout", "error", "precision", "exchange", "rccl", "workload", "sample"}


def _sig(x, n=6):
    return float(f"{x:.{n}g}") if isinstance(x, float) and x == x and abs(x) != float("inf") else x


def _shrink(x, key=""):
    """numbers rounded to 6 significant digits; strings over 72 characters are dropped unless their key is a name the
    contract asks for, in which case they are cut to 120 (`error` strings to 300)"""
    if isinstance(x, dict):
        out = {}
        for k, v in x.items():
            if isinstance(v, str) and len(v) > 72:
                if k not in _KEEP_STR:
                    continue
                v = v[:300 if k == "error" else 120]
            if k in ("traceback", "repeats_note", "speedup_note", "six_x_budget"):
                continue
            out[k] = _shrink(v, k)
        return out
    if isinstance(x, (list, tuple)):
        return [_shrink(v, key) for v in x]
    return _sig(x)


def _leg(e):
    """one `extra` leg -> {"v": value, "ms": ms per step, "frac": roofline.frac, "B": batch, ...}"""
    if not isinstance(e, dict) or "error" in e:
        return _shrink(e)
    ms = e.get("ms_per_step", e.get("us_per_step", 0.0) / 1e3 if "us_per_step" in e else None)
    out = {"v": e.get("value"), "unit": e.get("unit"), "B": e.get("batch"), "steps": e.get("steps"), "ms": ms}
    rf = e.get("roofline")
    if isinstance(rf, dict):
        out.update(bound=rf.get("bound"), frac=rf.get("frac"), achieved=rf.get("achieved"), peak=rf.get("peak"))
    if "points" in e:           # per point: [batch, precision / tag, value, ms per step, bound, frac, dispatches per step, MB per step]
        out["points"] = [[q.get("batch"), q.get("precision", q.get("tag")), q.get("value"), q.get("ms_per_step"),
                          (q.get("roofline") or {}).get("bound"), (q.get("roofline") or {}).get("frac"),
                          q.get("dispatches_per_step"), q.get("algorithmic_MB_per_step")] for q in e["points"]]
    for k in ("spmm", "wire_bytes_per_step", "small_batch"):
        if k in e:
            out[k] = e[k]
    cb = e.get("cpu_baseline")
    if isinstance(cb, dict):
        out["cpu"] = {k: cb.get(k) for k in ("value", "unit", "cores", "kind", "batch", "steps") if k in cb}
    return _shrink({k: v for k, v in out.items() if v is not None})


def emit(out):
    """print the ONE JSON line (compact); keep the verbose record beside it"""
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "bench_full.json"), "w") as f:
            f.write(json.dumps(out) + "\n")
    except OSError:
        pass
    special = ("extra", "secondary", "cpu_baseline", "repeats")
    line = _shrink({k: v for k, v in out.items() if k not in special})
    if "repeats" in out and isinstance(line.get("roofline"), dict):
        line["roofline"]["repeats"] = _shrink(out["repeats"])
    if "extra" in out:           # the legs before the two objects the contract asks for (a cut tail loses the least)
        line["extra"] = {n: _leg(e) for n, e in out["extra"].items()}
    tail = {k: _shrink(out[k]) for k in ("secondary", "cpu_baseline") if k in out}
    line.update(tail)
    txt = json.dumps(line, separators=(",", ":"))
    line["line_bytes"] = len(txt)
    print(json.dumps(line, separators=(",", ":")), flush=True)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="default: two epochs of full batches")
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--batch", type=int, default=None,
                    help="interactions per GPU per step (default 2M at N=1, 16M per rank at N>1: DESIGN.md section 5)")
    ap.add_argument("--workload", default="auto", choices=["auto", "c2", "c3", "tiny"])
    ap.add_argument("--nnz", type=int, default=None,
                    help="override the workload's interaction count (the table shapes U, I and the batch stay: a shorter "
                         "epoch of the same per-step regime)")
    ap.add_argument("--item-mode", default="fused", choices=["fused", "chunked", "atomic", "sorted"])
    ap.add_argument("--plan", default="auto", choices=["auto", "indexed", "sorted"],
                    help="epoch plan layout: indexed = partitioned (staged step only), sorted = radix-sorted")
    ap.add_argument("--dist", default="uniform", choices=["uniform", "zipf"])
    ap.add_argument("--reg", type=float, default=0.001)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="N=1: skip the configs[2]-shapes leg")
    ap.add_argument("--no-ref", action="store_true", help="N>1: skip the same-workload 1-GPU measurement")
    ap.add_argument("--no-sweep", action="store_true", help="N>1: only the primary (B_local, slices) point")
    ap.add_argument("--no-replica-check", action="store_true", help="N>1: skip the sharded-fit vs single-GPU-fit check")
    ap.add_argument("--overlap-plan", type=int, default=0,
                    help="build the next epoch's plan on a side stream (two plans in ping-pong; measured: no gain, the "
                         "plan build and the steps compete for the same memory system)")
    ap.add_argument("--cpu-steps", type=int, default=3,
                    help="timed CPU steps at the headline batch (a step of 2M interactions costs the host ~7 s); the "
                         ">= 30 steps of SURVEY 8(d) run at B = 65536 on the same tables")
    ap.add_argument("--repeats", type=int, default=3,
                    help="timed regions of --steps steps each on the headline leg; `value` is their median (boxes and "
                         "passes differ by a few per cent)")
    ap.add_argument("--no-extras", action="store_true",
                    help="N=1: skip the `extra` legs (Zipf ids, Adam, B = 65536, NeuMF at ml-1m shapes, LightGCN at "
                         "Amazon-Book shapes)")
    ap.add_argument("--slices", type=int, default=0,
                    help="N>1: cut the item pass into this many item ranges and exchange a finished range on a side "
                         "stream while the next one is reduced (1: one exchange after the whole pass; 0 = automatic: "
                         "sharding.auto_exchange_slices, from the exchange bytes, an assumed bus rate and the batch)")
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


def cpu_baseline(U, I, d, B, batches, reg):
    """The reference's CPU/PyTorch path (oracle/torch_port.py restates it with the same stock ops; the
    reference checkout does not exist on the GPU box) timed on this host's cores on a bounded sample of the GPU run's
    own epoch (same triples): len(batches) steps at the headline batch B; the >= 30 steps SURVEY 8(d) asks for at
    B = 65536 (rows of the same batches), where a step costs the host a quarter of a second instead of seven; and, as
    context, the reference's DEFAULT batch (B = 256, basic.yaml:23) on the same tables."""
    from oracle.torch_port import TorchMFBPR
    torch.manual_seed(2022)
    m = TorchMFBPR(U, I, d, 0.01, reg, reg)
    u, i, j = batches[0]

    def leg(sb, steps):
        steps = max(1, min(steps, len(u) // sb - 1))          # slices [sb*(k+1), sb*(k+2)) must exist
        m.step(u[:sb], i[:sb], j[:sb])                        # warm-up (allocations, thread pool)
        t0 = time.perf_counter()
        for k in range(steps):
            sl = slice(sb * (k + 1), sb * (k + 2)) if len(u) >= 2 * sb else slice(0, sb)
            m.step(u[sl], i[sl], j[sl])
        dt = time.perf_counter() - t0
        return {"value": steps * sb / dt, "unit": "interactions/s", "batch": sb, "steps": steps, "seconds": dt}

    mid = leg(min(65536, len(u)), 30)                         # also warms the pool up for the big steps
    t0 = time.perf_counter()
    for b in batches:
        m.step(*b)
    dt = time.perf_counter() - t0
    steps = len(batches)
    small = leg(min(256, len(u)), 10)
    return {"value": steps * B / dt, "unit": "interactions/s", "cores": torch.get_num_threads(),
            "kind": "port",
            "sample": f"{steps} SGD steps at B={B} on the first batches of the GPU run's epoch (U={U}, I={I}, d={d}; "
                      f"oracle/torch_port.py: nn.Embedding + autograd + optim.SGD, dense grads like the reference), "
                      f"{dt:.1f}s",
            "why_not_30_steps_at_this_batch": "a step of this batch costs the host ~7 s (dense gradients over both "
                                              "tables) and the bench contract bounds the CPU leg to tens of seconds; "
                                              "the 30 steps of SURVEY 8(d) are in `steps30_b65536` (same tables, "
                                              "same triples), the rate is flat from the second step on",
            "steps30_b65536": mid,
            "reference_default_batch": small}


def build_data(a, rank, world, dev, wl, nnz_override=None):
    """tables, triples and the per-fit index of `wl` for (rank, world), resident in HBM"""
    from daisyrec_amd import ops
    d = 64
    if wl == "c2":
        U_loc, I, nnz_loc, scaling = 1_000_000, 100_000, 50_000_000, "weak"
        name = ("BASELINE configs[1]: MF+BPR synthetic 1M users x 100K items x 50M nnz, d=64" if world == 1 else
                f"weak scaling of BASELINE configs[1]: {world}M users x 100K items x {50 * world}M nnz, d=64, "
                f"user-sharded over {world} GPUs (1M users / 50M nnz per GPU)")
    elif wl == "c3":
        U_tot, I, nnz_tot, scaling = 10_000_000, 1_000_000, 500_000_000, "strong"
        U_loc, nnz_loc = U_tot // world, nnz_tot // world
        name = f"BASELINE configs[2]: MF+BPR synthetic 10M users x 1M items x 500M nnz, d=64, user-sharded over {world} GPU(s)"
    else:
        U_loc, I, nnz_loc, scaling = 20_000, 5_000, 1_000_000, "weak"
        name = "tiny smoke workload (NOT a BASELINE config)"
    nnz_cut = nnz_override if nnz_override is not None else a.nnz
    if nnz_cut is not None:
        nnz_loc = nnz_cut // world
        name += f" [interactions cut to {nnz_cut}: same tables and batch, shorter epoch]"
    triples = synth_triples(U_loc, I, nnz_loc, 2022 + rank, dev, a.dist)      # LOCAL user ids
    g = torch.Generator(device=dev)
    g.manual_seed(2022)
    Q = torch.empty(I, d, device=dev).normal_(0.0, 0.01, generator=g)         # identical on every rank
    g.manual_seed(7 + rank)
    P = torch.empty(U_loc, d, device=dev).normal_(0.0, 0.01, generator=g)
    user_sorted = ops.triples_user_sorted(triples)      # synthetic triples are generated in CSR order
    plan_kind = a.plan if a.plan != "auto" else ("indexed" if a.item_mode == "fused" else "sorted")
    index = ops.TrainIndex(triples, U_loc, I, user_sorted=user_sorted) if plan_kind == "indexed" else None
    return {"name": name, "scaling": scaling, "d": d, "U": U_loc, "I": I, "n": triples.shape[0], "triples": triples,
            "P": P, "Q": Q, "index": index, "plan_kind": plan_kind, "user_sorted": user_sorted}


def free_data(data):
    if data.get("index") is not None:
        data["index"].close()
    for k in ("triples", "P", "Q", "index"):
        data.pop(k, None)
    torch.cuda.empty_cache()


def measure(a, data, rank, world, dev, B, slices, steps, warmup, want_cpu_batches=0, repeats=1, exchange="dense"):
    """warmup + `steps` timed steps of batch size B (per rank) over `data`; max over ranks of the wall time.
    repeats > 1: that many timed regions of `steps` steps, each bracketed like the first; the result is the region
    with the MEDIAN wall time, the others are listed in `repeats`."""
    from daisyrec_amd import ops
    from daisyrec_amd.sharding import UserShardedBprTrainer
    d, n, U_loc, I = data["d"], data["n"], data["U"], data["I"]
    triples, P, Q, index = data["triples"], data["P"], data["Q"], data["index"]
    B = min(B, n)
    lr, reg = 0.01, a.reg
    ctx = ops.BprContext(B, d, U_loc, I, device=dev)
    item_mode = ops.ITEM_MODES[a.item_mode]
    trainer = (UserShardedBprTrainer(ctx, P, Q, 0, lr, reg, reg, item_mode=item_mode, slices=slices or "auto", exchange=exchange)
               if world > 1 else None)
    if trainer is not None:
        slices = trainer.slices                # (0 / 'auto' resolved: the same on every rank by construction)
    if trainer is not None:
        trainer.enable_timing()
    full_batches = n // B                      # the bench steps over full batches only (fixed B per step)
    if world > 1:                              # shards differ by a few interactions (dedup): agree on the
        fb = torch.tensor([full_batches], device=dev, dtype=torch.int64)     # count, every rank must take the same steps
        dist.all_reduce(fb, op=dist.ReduceOp.MIN)
        full_batches = int(fb.cpu())
    assert full_batches >= 1, "batch larger than the rank's interaction count"
    steps = steps if steps is not None else 2 * full_batches

    plans = [ops.EpochPlan(n, U_loc, I, device=dev) for _ in range(2 if a.overlap_plan else 1)]
    side = torch.cuda.Stream(device=dev) if a.overlap_plan else None
    state = {"epoch": 0, "k": None, "cur": 0, "ready": None}

    def build_plan(slot, epoch):
        if index is not None:
            plans[slot].build_indexed(index, B, order="feistel", seed=2022 + rank, epoch=epoch)
        else:
            plans[slot].build(triples, B, order="feistel", seed=2022 + rank, epoch=epoch, user_sorted=data["user_sorted"])

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

    for _ in range(warmup):
        step()
    regions = []
    for rep in range(max(1, repeats)):
        # start the timed region on an epoch boundary: every timed epoch then contains exactly one plan build
        state["k"] = full_batches if (a.overlap_plan and state["k"] is not None) else None
        if trainer is not None and trainer.timeline is not None:
            trainer.timeline.clear()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        barrier()
        t0 = time.perf_counter()
        for k in range(steps):
            ev[k][0].record()
            step()
            ev[k][1].record()
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.cpu())
        regions.append((dt, sorted(s_.elapsed_time(e_) for s_, e_ in ev), trainer.step_split() if trainer is not None else None))
    loss_sum, nan_cnt = (float(x) for x in ctx.epoch_acc.cpu())
    assert nan_cnt == 0 and loss_sum == loss_sum, "NaN loss during the benchmark"
    order = sorted(range(len(regions)), key=lambda r_: regions[r_][0])
    dt, step_ms, split = regions[order[len(order) // 2]]
    all_dt = [r_[0] for r_ in regions]
    cpu_batches = []
    if want_cpu_batches:                       # the first batches of the last built epoch, for the CPU leg
        for k in range(min(want_cpu_batches, full_batches)):
            u, i, j = plans[state["cur"]].read_batch(k, B)[:3]
            cpu_batches.append(tuple(x.to(torch.int64).cpu() for x in (u, i, j)))
    res = {"name": data["name"], "scaling": data["scaling"], "B": B, "d": d, "n": n, "U": U_loc, "I": I, "steps": steps,
           "dt": dt, "lr": lr, "reg": reg, "step_ms": step_ms, "plan_kind": data["plan_kind"], "cpu_batches": cpu_batches,
           "plan_bytes": sum(p.nbytes for p in plans), "index_bytes": index.nbytes if index is not None else 0,
           "staged": trainer.staged if trainer is not None else (a.item_mode == "fused"), "slices": slices,
           "split_ms": split, "loss_sum": loss_sum, "all_dt": all_dt,
           "wire_bytes": dict(trainer.wire_bytes) if trainer is not None else None}
    ctx.close()
    for p in plans:
        p.close()
    return res


def roofline_of(r, world):
    """achieved algorithmic GB/s per GPU of a measurement (at N=1 never better than the wall clock)"""
    step_ms = r["step_ms"]
    gpu_ms_mean = sum(step_ms) / len(step_ms)
    eff_ms = max(gpu_ms_mean, r["dt"] / r["steps"] * 1e3) if world == 1 else gpu_ms_mean
    achieved = ALGO_BYTES_PER_INTERACTION_SGD(r["d"]) * r["B"] / (eff_ms * 1e-3) / 1e9
    return achieved, gpu_ms_mean


ADAM_BYTES_PER_INTERACTION = lambda d: 72 * d + 12        # + m, v read and written for the three rows
MFMA_PEAK_TF = {"fp32": 157.3, "bf16": 2500.0}            # MI355X_MICROARCH.md: dense MFMA peaks (no sparsity)


def _timed(fn, reps):
    """(wall seconds, HIP-event ms) of `reps` calls of fn, bracketed by synchronisations"""
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return time.perf_counter() - t0, e0.elapsed_time(e1)


def _hbm_roof(bytes_per_unit, units, seconds, note=None):
    ach = bytes_per_unit * units / seconds / 1e9
    r = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
         "traffic": None, "algorithmic_bytes_per_interaction": bytes_per_unit}
    if note:
        r["note"] = note
    return r


def extra_mf_epoch_loop(data, dev, B, reg, cpu_ref=None):
    """SURVEY 8(d)'s mid-size batch: whole epochs through daisy_bpr_fit_epoch_sgd (the C loop `MF.fit` runs: one
    enqueue per epoch), the plan build of every epoch inside the timed region"""
    from daisyrec_amd import ops
    n = min(data["n"], B * 800) // B * B
    tr = data["triples"][:n].contiguous()
    index = ops.TrainIndex(tr, data["U"], data["I"], user_sorted=True)
    plan = ops.EpochPlan(n, data["U"], data["I"], device=dev)
    ctx = ops.BprContext(B, data["d"], data["U"], data["I"], device=dev)
    P, Q = data["P"].clone(), data["Q"].clone()
    ep = [0]

    def epoch():
        ep[0] += 1
        plan.build_indexed(index, B, order="feistel", seed=2022, epoch=ep[0])
        ctx.fit_epoch_sgd(plan, P, Q, 0.01, reg, reg, item_mode=ops.ITEM_MODES["fused"])

    epoch()
    wall, ms = _timed(epoch, 2)
    loss, bad = (float(x) for x in ctx.epoch_acc.cpu())
    assert bad == 0 and loss == loss
    nb = n // B
    ctx.close(); plan.close(); index.close()
    out = {"workload": f"MF+BPR SGD, configs[1] tables, B = {B} through daisy_bpr_fit_epoch_sgd ({nb} steps per epoch, "
                       "plan builds timed)", "batch": B, "steps": 2 * nb, "value": 2 * n / wall, "unit": "interactions/s",
           "us_per_step": wall / (2 * nb) * 1e6, "gpu_us_per_step_events": ms / (2 * nb) * 1e3,
           "roofline": _hbm_roof(ALGO_BYTES_PER_INTERACTION_SGD(data["d"]), 2 * n, wall)}
    if cpu_ref is not None:
        out["cpu_baseline"] = dict(cpu_ref, cores=torch.get_num_threads(), kind="port",
                                   sample=f"{cpu_ref['steps']} SGD steps at B = {cpu_ref['batch']} on the same tables "
                                          "(oracle/torch_port.py)")
    return out


def extra_mf_adam(data, dev, B, reg):
    """MF + torch.optim.Adam (AbstractRecommender.py:54): the staged step whose row owners apply the exact lazy form,
    one epoch per enqueue (daisy_bpr_fit_epoch_adam), flush and plan build inside the timed region; priced by SURVEY
    8(d)'s Adam byte model (72 d + 12 B per interaction)"""
    from daisyrec_amd import ops
    n, U, I, d = data["n"], data["U"], data["I"], data["d"]
    plan = ops.EpochPlan(n, U, I, device=dev)
    ctx = ops.BprContext(B, d, U, I, device=dev)
    P, Q = data["P"].clone(), data["Q"].clone()
    nb = n // B
    adam = ops.LazyAdam(P, Q, 0.001, 4 * (nb + 1))
    ep = [0]

    def epoch():
        ep[0] += 1
        plan.build_indexed(data["index"], B, order="feistel", seed=2022, epoch=ep[0])
        adam.fit_epoch(ctx, plan, reg, reg)

    epoch()
    wall, ms = _timed(epoch, 1)
    loss, bad = (float(x) for x in ctx.epoch_acc.cpu())
    assert bad == 0 and loss == loss
    steps = plan.num_batches
    ctx.close(); plan.close()
    del adam
    return {"workload": f"MF+BPR with torch.optim.Adam semantics (lazy rows, exact), tables {U} x {I}, B = {B}, one epoch "
                        f"= {steps} steps + flush + plan build", "batch": B, "steps": steps, "value": n / wall,
            "unit": "interactions/s", "ms_per_step": wall / steps * 1e3, "gpu_ms_per_step_events": ms / steps,
            "roofline": _hbm_roof(ADAM_BYTES_PER_INTERACTION(d), n, wall,
                                  "SURVEY 8(d): 72 d + 12 B per interaction (rows + both moments read and written), no "
                                  "credit for duplicate rows - which is generous where a batch holds several samples per "
                                  "user and item (configs[1] tables at B = 2 M: 2.3 per touched user, 42 per item; the lazy "
                                  "form moves a touched row's moments once): read the figure at configs[2] table shapes "
                                  "(`mf_adam_c3shapes`) as the honest one")}


def extra_neumf(dev, want_cpu):
    """BASELINE configs[3]: NeuMF (GMF + 3-layer MLP) at ml-1m shapes, d = 64; a step = daisy_neumf_step_grads + one
    dense Adam pass.  The tower is GEMM work: priced against the dense MFMA peak of the mode's input type (fp32 = the
    reference's arithmetic and the parity mode; bf16 storage = the throughput mode configs[3] names)"""
    from daisyrec_amd import ops
    U, I, D, L = 6040, 3706, 64, 3
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    dm = D << (L - 1)
    shapes = {"uG": (U, D), "iG": (I, D), "uM": (U, dm), "iM": (I, dm)}
    w, macs = 2 * dm, 0
    for l in range(1, L + 1):
        shapes[f"W{l}"], shapes[f"b{l}"] = (w // 2, w), (w // 2,)
        macs += w * (w // 2)
        w //= 2
    shapes["Wp"], shapes["bp"] = (1, 2 * D), (1,)
    names = ops.neumf_param_names(L)
    numel = lambda k: int(torch.tensor(shapes[k]).prod())     # noqa: E731
    flat = torch.randn(sum(numel(k) for k in names), device=dev, generator=g) * 0.05
    gflat, m, v = torch.zeros_like(flat), torch.zeros_like(flat), torch.zeros_like(flat)
    p, gr, off = {}, {}, 0
    for k in names:
        p[k], gr[k] = flat[off:off + numel(k)].view(shapes[k]), gflat[off:off + numel(k)].view(shapes[k])
        off += numel(k)
    flops_per_sample = 2 * macs * 3 * 2            # x2 flop/MAC, x3 GEMMs (forward, d-input, d-weight), x2 rows (pos, neg)
    # bf16 storage, dropout 0, fewer distinct table rows than rows per step: the first layer runs through the embedding
    # tables (csrc/neumf.hip "FACT": two GEMMs over the U + I table rows instead of three over the 2 B rows of the step)
    macs_l1 = (2 * dm) * dm
    table_flops = 2 * 3 * (U + I) * dm * dm        # T = table x W1[:, half]^T, g.table += S W1[:, half], gW1 += S^T table
    out = {"workload": "BASELINE configs[3] shapes: NeuMF (GMF + 512-256-128-64 MLP) on ml-1m sizes U=6040, I=3706, d=64, "
                       "pairwise BPR rows (u, i, j), Adam; synthetic ids, random-init weights", "points": []}
    for B, prec, steps in ((65536, 0, 8), (262144, 0, 4), (65536, 2, 8), (262144, 2, 6)):
        u, i, j = (torch.randint(0, hi, (B,), device=dev, generator=g, dtype=torch.int32) for hi in (U, I, I))
        ctx = ops.NeumfContext(2 * B, D, L, U, I)
        ctx.set_precision(prec)
        t = [0]

        def step():
            t[0] += 1
            ctx.step_grads(p, gr, u, i, j, 0, 1e-3, 1e-3, dropout=0.0, seed=t[0])
            ops.adam_dense(flat, gflat, m, v, 1e-3, t[0])

        for _ in range(2):
            step()
        wall, ms = _timed(step, steps)
        ctx.close()
        name = "fp32" if prec == 0 else "bf16"
        fact = U + I <= 2 * B                      # both modes run the first layer through the tables then (round 6: fp32 too)
        step_flops = (flops_per_sample - (2 * macs_l1 * 3 * 2 if fact else 0)) * B + (table_flops if fact else 0)
        tf = step_flops * steps / wall / 1e12
        pt = {"batch": B, "precision": {0: "fp32", 2: "bf16"}[prec], "steps": steps, "value": B * steps / wall, "unit": "samples/s",
              "ms_per_step": wall / steps * 1e3, "gpu_ms_per_step_events": ms / steps, "first_layer_through_the_tables": fact,
              "mfma_tflops_of_the_gemms_run": tf, "nominal_tflops_of_the_plain_formulation": flops_per_sample * B * steps / wall / 1e12}
        if prec == 2 and fact:
            # the fused tower (csrc/neumf_tower.hip) leaves ~0.13 TFLOP per step for the matrix cores (50 us at the bf16
            # peak): the step is bound by what it moves.  Algorithmic bytes per row of the step (2 rows per sample), DESIGN.md
            # section 9: the two table-product rows (bf16, 2 n1 B each) + the two GMF rows (fp32, 4 d B each) gathered, dZ1
            # (bf16, 2 n1 B) written once and read once per table side by the two segmented sums, the other table's GMF row
            # gathered once per side, 12 B of ids - no credit for rows that repeat inside a step
            n1 = dm
            bytes_row = 2 * 2 * n1 + 2 * 4 * D + 2 * n1 + 2 * 2 * n1 + 2 * 4 * D + 12
            gbs = bytes_row * 2 * B * steps / wall / 1e9
            pt["algorithmic_MB_per_step"] = bytes_row * 2 * B / 1e6
            pt["roofline"] = {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                              "traffic": None, "algorithmic_bytes_per_row": bytes_row}
        else:
            pt["roofline"] = {"bound": "mfma", "achieved": tf, "peak": MFMA_PEAK_TF[name], "unit": "TFLOP/s",
                              "frac": tf / MFMA_PEAK_TF[name], "traffic": None}
        out["points"].append(pt)
    best = max(out["points"], key=lambda q: q["value"])
    out["value"], out["unit"] = best["value"], "samples/s"
    if want_cpu:
        from oracle.torch_port import TorchNeuMF
        torch.manual_seed(0)
        mdl = TorchNeuMF(U, I, D, L)
        gh = torch.Generator()
        gh.manual_seed(1)
        Bc, cs = 16384, 3
        bt = [tuple(torch.randint(0, hi, (Bc,), generator=gh) for hi in (U, I, I)) for _ in range(cs + 1)]
        mdl.step(*bt[0])
        t0 = time.perf_counter()
        for b in bt[1:]:
            mdl.step(*b)
        dt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": cs * Bc / dt, "unit": "samples/s", "cores": torch.get_num_threads(), "kind": "port",
                               "sample": f"{cs} Adam steps at B = {Bc} (oracle/torch_port.py: TorchNeuMF, stock "
                                         f"nn.Embedding / Linear / autograd / optim.Adam), {dt:.1f}s"}
    return out


def extra_small_batch(dev):
    """The reference's default batch (basic.yaml:23: B = 256) at BASELINE configs[0] sizes (ml-100k after the 10-filter: 943
    users x 1152 items, 78 363 triples, d = 32): every step of an epoch inside one persistent workgroup (csrc/bpr_small.hip) -
    SGD (MF's default, mf.yaml) and torch.optim.Adam in its exact lazy form (round 6).  The bound here is the dependency
    chain of a step, not bytes: priced in us per step, the HBM fraction is reported for completeness."""
    from daisyrec_amd import ops
    U, I, nnz, d, B = 943, 1152, 78363, 32, 256
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    u = torch.sort(torch.randint(0, U, (nnz,), device=dev, generator=g)).values
    tri = torch.stack([u, torch.randint(0, I, (nnz,), device=dev, generator=g),
                       torch.randint(0, I, (nnz,), device=dev, generator=g)], 1).to(torch.int32).contiguous()
    plan = ops.EpochPlan(nnz, U, I, device=dev).build(tri, B, order="feistel", seed=1, epoch=0, user_sorted=True)
    nb = plan.num_batches
    out = {"workload": "BASELINE configs[0] sizes (ml-100k: 943 x 1152, 78 363 triples, d = 32), B = 256, one persistent workgroup per epoch",
           "points": []}
    for opt in ("sgd", "adam"):
        P = torch.empty(U, d, device=dev).normal_(0.0, 0.01, generator=g)
        Q = torch.empty(I, d, device=dev).normal_(0.0, 0.01, generator=g)
        ctx = ops.BprContext(B, d, U, I, device=dev)
        adam = ops.LazyAdam(P, Q, 0.001, 8 * nb) if opt == "adam" else None

        def epoch():
            if adam is not None:
                adam.fit_epoch(ctx, plan, 1e-3, 1e-3)
            else:
                ctx.fit_epoch_sgd(plan, P, Q, 0.01, 1e-3, 1e-3)

        epoch()
        wall, ms = _timed(epoch, 3)
        ctx.close()
        per_step = wall / (3 * nb)
        bytes_i = ALGO_BYTES_PER_INTERACTION_SGD(d) if opt == "sgd" else 72 * d + 12
        gbs = bytes_i * B / per_step / 1e9
        out["points"].append({"batch": B, "tag": opt, "steps": 3 * nb, "value": B / per_step, "unit": "interactions/s",
                              "ms_per_step": per_step * 1e3,
                              "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                           "frac": gbs / HBM_PEAK_GBS, "traffic": None}})
    plan.close()
    out["value"], out["unit"] = max(q["value"] for q in out["points"]), "interactions/s"
    # NeuMF at the reference's own operating point (neumf.yaml: factors 24, 2 layers, dropout 0.5, Adam; basic.yaml:23: B = 256)
    # through NeuMF.fit: gather ... input gradient in one launch with the weights in LDS (csrc/neumf_mid.hip), 4 dispatches per
    # step, the epoch's loop issued by the library (daisy_neumf_fit_epoch)
    import logging
    import numpy as np
    from daisyrec_amd.model.NeuMFRecommender import NeuMF
    from daisyrec_amd.utils.dataset import BasicDataset, get_dataloader
    n = B * 300
    rng = np.random.default_rng(0)
    tri_n = np.stack([rng.integers(0, U, n), rng.integers(0, I, n), rng.integers(0, I, n)], 1).astype(np.int32)
    cfg = {"gpu": str(dev.index or 0), "logger": logging.getLogger("bench"), "lr": 0.001, "reg_1": 0.0, "reg_2": 0.001, "epochs": 1,
           "topk": 50, "user_num": U, "item_num": I, "factors": 24, "num_layers": 2, "dropout": 0.5, "loss_type": "BPR",
           "optimizer": "adam", "init_method": "default", "early_stop": False, "model_name": "NeuMF", "GMF_model": None,
           "MLP_model": None, "algo_name": "neumf", "progress": False}
    model = NeuMF(cfg)
    loader = get_dataloader(BasicDataset(tri_n), batch_size=B, shuffle=False, num_workers=0)
    model.fit(loader)
    wall, _ = _timed(lambda: model.fit(loader), 2)
    per_step = wall / (2 * 300)
    out["points"].append({"batch": B, "tag": "neumf_defaults_adam", "steps": 600, "value": B / per_step, "unit": "samples/s",
                          "ms_per_step": per_step * 1e3, "dispatches_per_step": 4,
                          "workload": "NeuMF.fit, neumf.yaml defaults (factors 24, 2 layers, dropout 0.5, Adam), ml-100k sizes"})
    return out


def extra_lightgcn(dev, want_cpu):
    """BASELINE configs[4] shapes on ONE GPU: LightGCN, 3 layers, Amazon-Book sizes; the sparse x dense products are
    HBM-bound (per stored entry one 4 d-byte row gather + 20 B of entry metadata, per node a memset and a row write)"""
    import logging
    import numpy as np
    import scipy.sparse as sp
    from daisyrec_amd import ops
    from daisyrec_amd.model.LightGCNRecommender import LightGCN
    U, I, NNZ, D, L = 52643, 91599, 2380730, 64, 3
    rng = np.random.default_rng(0)
    w = 1.0 / np.arange(1, I + 1) ** 0.8                       # popularity-skewed items
    cdf = np.cumsum(w / w.sum())
    gi = np.minimum(np.searchsorted(cdf, rng.random(NNZ)), I - 1).astype(np.int64)
    gu = rng.integers(0, U, NNZ).astype(np.int64)
    cfg = dict(gpu="0", logger=logging.getLogger("bench"), epochs=1, lr=0.01, topk=50, user_num=U, item_num=I,
               inter_matrix=sp.coo_matrix((np.ones(NNZ, np.float32), (gu, gi)), shape=(U, I)), factors=D,
               num_layers=L, reg_1=0.0, reg_2=0.0, loss_type="BPR", optimizer="default", init_method="default",
               early_stop=False, progress=False)
    torch.manual_seed(0)
    model = LightGCN(cfg)
    E0 = model._ego()
    graph = model._adj()
    X = torch.randn(U + I, D, device=dev)
    graph.spmm(X)
    wall, _ = _timed(lambda: graph.spmm(X), 20)
    spmm_s = wall / 20
    alg = graph.nnz * (4 * D + 20) + (U + I) * 4 * D * 2
    out = {"workload": f"BASELINE configs[4] shapes on one GPU: LightGCN {L} layers, Amazon-Book sizes U={U}, I={I}, "
                       f"{NNZ} interactions -> {graph.nnz} stored adjacency entries, d={D}; the propagation is "
                       "recomputed for every batch like the reference (LightGCNRecommender.py:117-129,141)",
           "spmm": {"ms": spmm_s * 1e3, "G_row_gathers_per_s": graph.nnz / spmm_s / 1e9,
                    "roofline": {"bound": "hbm", "achieved": alg / spmm_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": alg / spmm_s / 1e9 / HBM_PEAK_GBS, "traffic": None,
                                 "algorithmic_bytes": alg,
                                 "note": "the 37 MB table is Infinity-Cache resident: cache-assisted"}},
           "points": []}
    loss_id = ops.loss_id("BPR")
    o_, G, dE0 = torch.empty_like(E0), torch.empty_like(E0), torch.zeros_like(E0)
    m, v = torch.zeros_like(model._flat), torch.zeros_like(model._flat)
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    for B, steps in ((4096, 20), (65536, 20)):
        u, i, j = (torch.randint(0, hi, (B,), device=dev, generator=g, dtype=torch.int32) for hi in (U, I, I))
        ctx = ops.BprContext(B, D, U, I)
        t = [0]

        def step():
            t[0] += 1
            model._batch_grads(ctx, E0, o_, G, dE0, u, i, j, loss_id)
            ops.adam_dense(model._flat, dE0.view(-1), m, v, 0.01, t[0])

        for _ in range(2):
            step()
        wall, ms = _timed(step, steps)
        ctx.close()
        out["points"].append({"batch": B, "steps": steps, "value": B * steps / wall, "unit": "samples/s",
                              "ms_per_step": wall / steps * 1e3, "gpu_ms_per_step_events": ms / steps,
                              "propagation_share": 2 * L * spmm_s / (wall / steps),
                              "roofline": {"bound": "hbm", "achieved": 2 * L * alg / (wall / steps) / 1e9, "peak": HBM_PEAK_GBS,
                                           "unit": "GB/s", "frac": 2 * L * alg / (wall / steps) / 1e9 / HBM_PEAK_GBS,
                                           "traffic": None,
                                           "note": "algorithmic bytes of the step's 2 L products only (the batch part and "
                                                   "the dense Adam pass are not counted) / the whole step's time"}})
    out["value"], out["unit"] = out["points"][-1]["value"], "samples/s"
    if want_cpu:
        from oracle import lightgcn_numpy as LG
        from oracle.torch_port import TorchLightGCN
        indptr, col, val = LG.norm_adj_csr(gu, gi, U, I)
        rows = np.repeat(np.arange(U + I), np.diff(indptr))
        adj = torch.sparse_coo_tensor(torch.as_tensor(np.stack([rows, col.astype(np.int64)])), torch.as_tensor(val),
                                      (U + I, U + I)).coalesce()
        torch.manual_seed(0)
        mdl = TorchLightGCN(U, I, D, L, adj)
        gh = torch.Generator()
        gh.manual_seed(1)
        Bc, cs = 4096, 2
        bt = [tuple(torch.randint(0, hi, (Bc,), generator=gh) for hi in (U, I, I)) for _ in range(cs + 1)]
        mdl.step(*bt[0])
        t0 = time.perf_counter()
        for b in bt[1:]:
            mdl.step(*b)
        dt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": cs * Bc / dt, "unit": "samples/s", "ms_per_step": dt / cs * 1e3,
                               "cores": torch.get_num_threads(), "kind": "port",
                               "sample": f"{cs} Adam steps at B = {Bc} (oracle/torch_port.py: TorchLightGCN, "
                                         f"torch.sparse.mm propagation per batch), {dt:.1f}s"}
    graph.close()
    return out


def replica_check(a, rank, world, dev):
    """A small fit through the reference's API - `MF.fit(train_loader)` on every rank, which shards the users over the
    ranks by itself (reduce-scatter / owner apply / all-gather per step over the job's backend) - against the SAME fit
    on one GPU (rank 0, sharding switched off): the epoch losses must agree to 1e-6 relative, Q must be identical on
    all ranks (its checksums are all-gathered) and within fp32 summation-order distance of the single-GPU Q."""
    import logging
    import numpy as np
    from daisyrec_amd.model.MFRecommender import MF
    from daisyrec_amd.utils.dataset import BasicDataset, get_dataloader
    U, I, n, B, d = 40_000, 8_000, 600_000, 65_536, 64
    rng = np.random.default_rng(11)
    tri = np.stack([np.sort(rng.integers(0, U, n)), rng.integers(0, I, n), rng.integers(0, I, n)], 1).astype(np.int32)
    cfg = {"gpu": "0", "logger": logging.getLogger("bench"), "lr": 0.01, "reg_1": 0.001, "reg_2": 0.001, "epochs": 2,
           "topk": 10, "user_num": U, "item_num": I, "factors": d, "loss_type": "BPR", "optimizer": "sgd",
           "init_method": "default", "early_stop": False, "shuffle_mode": "device", "progress": False, "seed": 7}

    def fit(shard):
        torch.manual_seed(123)
        m = MF(dict(cfg, shard_users=shard))
        m.fit(get_dataloader(BasicDataset(tri), batch_size=B, shuffle=True, num_workers=0))
        return m

    t0 = time.perf_counter()
    m = fit(True)
    Q = m.embed_item.weight.data
    chk = torch.stack([Q.double().sum(), Q.double().abs().sum(), Q.view(torch.int32).to(torch.int64).sum().double()])
    all_chk = [torch.zeros_like(chk) for _ in range(world)]
    dist.all_gather(all_chk, chk)
    same = all(bool(torch.equal(c, all_chk[0])) for c in all_chk)
    out = {"workload": f"MF.fit sharded over {world} ranks vs one GPU: U={U}, I={I}, n={n}, B={B}, d={d}, 2 epochs, device shuffle",
           "q_checksum_identical_on_all_ranks": same, "epoch_losses_sharded": [float(x) for x in m.epoch_losses]}
    if rank == 0:
        ref = fit(False)
        Qr = ref.embed_item.weight.data
        rel = max(abs(x - y) / abs(y) for x, y in zip(m.epoch_losses, ref.epoch_losses))
        dq = (Q - Qr).abs()
        dp = (m.embed_user.weight.data - ref.embed_user.weight.data).abs()
        out.update({"epoch_losses_1gpu": [float(x) for x in ref.epoch_losses], "loss_max_rel_diff_vs_1gpu": rel,
                    "loss_equal_to_1e-6": bool(rel <= 1e-6),
                    # fp32 summation order differs between the two runs (typically 1e-7 per element); an element that
                    # crosses zero sees the L1 regulariser's sign flip one step apart: a jump of 2 * lr * reg_1 * count
                    "q_max_abs_diff_vs_1gpu": float(dq.max().cpu()), "p_max_abs_diff_vs_1gpu": float(dp.max().cpu()),
                    "q_fraction_beyond_1e-5": float((dq > 1e-5).double().mean().cpu()),
                    "p_fraction_beyond_1e-5": float((dp > 1e-5).double().mean().cpu()),
                    "seconds": time.perf_counter() - t0})
    dist.barrier()
    return out


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert torch.cuda.is_available(), "bench.py needs a HIP device (no CPU fallback)"
    ndev = torch.cuda.device_count()
    if world != a.gpus:
        if world > 1:
            raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
        # --gpus N without a launcher: spawn the N ranks here; never run a smaller job under the bigger name
        if a.backend == "nccl" and ndev < a.gpus:
            raise SystemExit(f"bench.py --gpus {a.gpus}: this box has {ndev} GPU(s); refusing to run a "
                             f"{ndev}-GPU job labelled n_gpus={a.gpus}")
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        import torch.multiprocessing as mp
        mp.spawn(_spawned, args=(a.gpus, port, sys.argv[1:]), nprocs=a.gpus, join=True)
        return
    if a.backend == "nccl" and world > ndev:
        raise SystemExit(f"bench.py: WORLD_SIZE={world} but only {ndev} GPU(s) are visible (one rank per GPU)")
    local_rank = local_rank % ndev
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    diag = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)   # RCCL
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        # who is here: every rank reports its device; a duplicate device or a short world shows on the line
        props = torch.cuda.get_device_properties(dev)
        me = {"rank": rank, "local_rank": local_rank, "device_index": dev.index, "name": props.name,
              "uuid": str(getattr(props, "uuid", "")), "pid": os.getpid()}
        everyone = [None] * world
        dist.all_gather_object(everyone, me)
        try:
            rccl = ".".join(str(x) for x in torch.cuda.nccl.version())
        except Exception as e:                          # noqa: BLE001
            rccl = f"unavailable ({e})"
        diag = {"backend": dist.get_backend(), "rccl_version": rccl, "world_size": dist.get_world_size(),
                "visible_devices": ndev, "ranks": everyone,
                "distinct_devices": len({(r["device_index"], r["uuid"]) for r in everyone}),
                "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")}

    wl = a.workload if a.workload != "auto" else ("c2" if world == 1 else "c3")
    want_cpu = a.cpu_steps if (world == 1 and rank == 0 and not a.no_cpu_baseline) else 0
    B_main = a.batch if a.batch is not None else ((1 << 21) if world == 1 else (1 << 24))

    def guarded(what, fn, *args, collective=True):
        """a diagnostic leg must not cost the run its headline number: an exception in it is put on the line instead.
        Collective legs: a rank that failed alone (an OOM in rank 0's extra work, say) would leave the others blocked in
        the leg's next collective if it simply carried on - it cannot rejoin them mid-leg either, so the ranks agree on
        the outcome AFTER the leg (the failing rank gets there because an exception left the leg; the healthy ones
        because RCCL's watchdog / the leg's end released them) and every rank reports the failure alike."""
        err = None
        try:
            res = fn(*args)
        except Exception as e:                          # noqa: BLE001
            import traceback
            err = {"error": f"{what}: {type(e).__name__}: {e}", "traceback": traceback.format_exc()[-1500:]}
            res = err
        if collective and world > 1:
            flag = torch.tensor([1 if err is not None else 0], device=dev, dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            if int(flag.cpu()) and err is None:
                res = {"error": f"{what}: failed on another rank"}
        return res

    check = None
    if world > 1 and not a.no_replica_check and a.item_mode == "fused":
        check = guarded("replica_check", replica_check, a, rank, world, dev)

    data = build_data(a, rank, world, dev, wl)
    main_note = None
    if world > 1 and a.slices != 1:
        # the pipelined exchange has never met more than one rank of RCCL: if the main point fails with it, every rank
        # agrees on that (guarded) and the point is measured with ONE exchange per step instead - the failure stays on the line
        r = guarded("main point with the pipelined exchange", measure, a, data, rank, world, dev, B_main, a.slices, a.steps,
                    a.warmup, want_cpu, a.repeats)
        if "error" in r:
            main_note = r
            torch.cuda.empty_cache()
            r = measure(a, data, rank, world, dev, B_main, 1, a.steps, a.warmup, want_cpu, repeats=a.repeats)
    else:
        r = measure(a, data, rank, world, dev, B_main, a.slices, a.steps, a.warmup, want_cpu, repeats=a.repeats)
    # N = 1: the other operating points of SURVEY 8(d) and the other BASELINE configs, on the same JSON line as `extra`
    # (each a guarded leg: an exception lands on the line instead of costing the headline number)
    want_extras = (world == 1 and wl == "c2" and not a.no_extras and a.item_mode == "fused" and a.batch is None
                   and a.nnz is None and a.dist == "uniform")
    extra = {}
    cpu_mid = [None]
    if want_extras:
        if r["cpu_batches"]:            # the CPU leg first: its 30-step B = 65536 figure pairs with the GPU leg at that batch
            cpu_res = guarded("cpu_baseline", cpu_baseline, r["U"], r["I"], r["d"], r["B"], r["cpu_batches"], r["reg"],
                              collective=False)
            r["cpu_result"] = cpu_res
            cpu_mid[0] = cpu_res.get("steps30_b65536") if isinstance(cpu_res, dict) else None
        extra["mf_b65536"] = guarded("extra mf_b65536", extra_mf_epoch_loop, data, dev, 65536, a.reg, cpu_mid[0],
                                     collective=False)
        extra["mf_adam"] = guarded("extra mf_adam", extra_mf_adam, data, dev, 1 << 21, a.reg, collective=False)

        def variant_leg(what, B_, reg_):
            av = argparse.Namespace(**vars(a))
            av.reg = reg_
            rv = measure(av, data, rank, world, dev, B_, 1, None, a.warmup)       # two whole epochs: one plan build per epoch
            achv, gpuv = roofline_of(rv, 1)
            return {"workload": f"configs[1], {what}", "batch": rv["B"], "steps": rv["steps"],
                    "value": rv["steps"] * rv["B"] / rv["dt"], "unit": "interactions/s",
                    "ms_per_step": rv["dt"] / rv["steps"] * 1e3, "gpu_ms_per_step_events": gpuv,
                    "roofline": {"bound": "hbm", "achieved": achv, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": achv / HBM_PEAK_GBS, "traffic": None,
                                 "algorithmic_bytes_per_interaction": ALGO_BYTES_PER_INTERACTION_SGD(64)}}
        extra["mf_reg0"] = guarded("extra mf_reg0", variant_leg, "reg_1 = reg_2 = 0 (SURVEY 8d's variant), B = 2097152", 1 << 21,
                                   0.0, collective=False)
        extra["mf_b1m"] = guarded("extra mf_b1m", variant_leg, "B = 1048576 (SURVEY 8d's 1 M per GPU)", 1 << 20, a.reg,
                                  collective=False)
    # N > 1: everything after the main point is diagnosis (replica-independent sweeps, exchange points, the one-GPU reference) and
    # most of it has never met more than one rank of RCCL.  If it has not finished within DAISY_BENCH_OPTIONAL_BUDGET_S
    # (default 600 s; about one minute is normal), rank 0 prints the contract's line from the main point alone and leaves - a
    # hang in a diagnostic leg must not cost the run its number.
    watchdog = None
    if (world > 1 or os.environ.get("DAISY_BENCH_FORCE_WATCHDOG")) and rank == 0:     # (the variable: a one-GPU test of this exit)
        import threading

        def headline_only():
            ach, gpu_ms = roofline_of(r, world)
            line = {"metric": "BPR training interactions/sec at d=64; achieved HBM GB/s vs peak",
                    "value": r["steps"] * r["B"] * world / r["dt"], "unit": "interactions/s", "n_gpus": world, "steps": r["steps"],
                    "warmup": a.warmup, "ms_per_step": r["dt"] / r["steps"] * 1e3, "higher_is_better": True,
                    "scaling": r["scaling"], "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                    "config": {"workload": r["name"], "batch_per_gpu": r["B"], "global_batch": r["B"] * world, "d": r["d"],
                               "optimizer": "sgd", "loss": "BPR", "parallelism": f"user-sharded dp{world}" if world > 1 else "single GPU",
                               "exchange_slices": r["slices"]},
                    "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                                 "traffic": None, "gpu_ms_per_step_events": gpu_ms},
                    "error": "the diagnostic legs after the main point did not finish in time: headline only"}
            emit(line)
            sys.stdout.flush()
            os._exit(0)

        watchdog = threading.Timer(float(os.environ.get("DAISY_BENCH_OPTIONAL_BUDGET_S", "600")), headline_only)
        watchdog.daemon = True
        watchdog.start()
    sweep = []
    if world > 1 and wl in ("c3", "tiny") and not a.no_sweep and a.item_mode == "fused" and (a.batch is None or wl == "tiny"):
        # the regimes of DESIGN.md section 5 in one launch: the exchange is a fixed 2 x 231 MB per step and rank, so
        # the batch decides how much compute it is spread over; slices > 1 hides it under the item pass
        # (the tiny workload runs a miniature of the same sweep: it exists to exercise this code on a one-GPU box)
        for Bl in ((1 << 21, 1 << 23, 1 << 24) if wl == "c3" else (1 << 14, 1 << 16)):
            for sl in ((1, 8) if wl == "c3" else (1, 4)):
                if (min(Bl, data["n"]), sl) == (r["B"], r["slices"]):
                    sweep.append(r)
                    continue
                fb = max(1, data["n"] // min(Bl, data["n"]))
                sweep.append(guarded(f"sweep point B_local={Bl} slices={sl}", measure, a, data, rank, world, dev, Bl, sl,
                                     min(fb, 16), 2))
    xchg = []
    if world > 1 and wl in ("c3", "tiny") and not a.no_sweep and a.item_mode == "fused" and (a.batch is None or wl == "tiny"):
        # the reference's batch sizes under the sharded step (basic.yaml:23: B = 256 ... 65 536): the dense exchange moves
        # the whole item table for a few thousand touched rows, the touched-rows exchange (sharding.py, round 6) only their
        # union - wire bytes per step and rank of both forms, and the measured step of each (no speed-up claim is made
        # for either: this is the first time they meet more than one rank)
        for Bl in ((4096, 32768) if wl == "c3" else (256, 2048)):
            for mode in ("dense", "sparse"):
                fb = max(1, data["n"] // min(Bl, data["n"]))
                rr = guarded(f"exchange point B_local={Bl} {mode}", measure, a, data, rank, world, dev, Bl, 1, min(fb, 16), 2,
                             0, 1, mode)
                xchg.append(rr if "error" in rr else
                            {"batch_per_gpu": rr["B"], "exchange": mode, "ms_per_step": rr["dt"] / rr["steps"] * 1e3,
                             "value": rr["steps"] * rr["B"] * world / rr["dt"], "wire_bytes_per_step_and_rank": rr["wire_bytes"][mode],
                             "auto_would_pick": __import__("daisyrec_amd.sharding", fromlist=["x"]).auto_item_exchange(
                                 data["I"], data["d"], world, rr["B"])})
    free_data(data)

    secondary = None
    if world == 1 and wl == "c2" and not a.no_secondary and a.item_mode == "fused" and a.batch is None and a.nnz is None:
        d2 = build_data(a, rank, world, dev, "c3", nnz_override=SECONDARY_NNZ)
        r2 = measure(a, d2, rank, world, dev, 1 << 21, 1, None, a.warmup)
        if want_extras:      # Adam where the byte model has no duplicate rows to be generous about: ~1.1 samples per touched user
            extra["mf_adam_c3shapes"] = guarded("extra mf_adam_c3shapes", extra_mf_adam, d2, dev, 1 << 21, a.reg, collective=False)
        free_data(d2)
        ach2, gpu2 = roofline_of(r2, 1)
        traffic2, traffic2_src = None, None
        t2path = os.path.join(ROOT, "profiles", "pmc_traffic_c3shapes.json")
        if os.path.exists(t2path):
            try:        # the committed PMC passes of this very leg (bench.py --workload c3 --nnz 100000000)
                t2 = json.load(open(t2path))
                traffic2 = t2["hbm_bytes_per_interaction"] * r2["B"]
                traffic2_src = t2.get("source", "profiles/pmc_traffic_c3shapes.json")
            except Exception:
                traffic2 = None
        secondary = {"workload": r2["name"] + " - the pure-HBM regime of configs[2] on ONE GPU (P 2.56 GB, Q 256 MB: "
                                              "both beyond the 256 MB Infinity Cache)",
                     "value": r2["steps"] * r2["B"] / r2["dt"], "unit": "interactions/s", "steps": r2["steps"],
                     "batch": r2["B"], "ms_per_step": r2["dt"] / r2["steps"] * 1e3, "gpu_ms_per_step_events": gpu2,
                     "roofline": {"bound": "hbm", "achieved": ach2, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": ach2 / HBM_PEAK_GBS, "traffic": traffic2, "traffic_source": traffic2_src,
                                  "algorithmic_bytes_per_interaction": ALGO_BYTES_PER_INTERACTION_SGD(64),
                                  "ceiling_note": "at these table shapes a step really moves ~1.38x the algorithmic bytes "
                                                  "(every P row read and written once per sample, a quarter of the Q rows "
                                                  "updated in place per step, the stage written once and read twice): at the "
                                                  "~6.3 TB/s the fabric delivers frac cannot exceed ~0.57 with this design"}}

    if want_extras:
        def zipf_leg():
            az = argparse.Namespace(**vars(a))
            az.dist = "zipf"
            dz = build_data(az, rank, world, dev, "c2")
            try:
                rz = measure(az, dz, rank, world, dev, 1 << 21, 1, None, a.warmup)
            finally:
                free_data(dz)
            achz, gpuz = roofline_of(rz, 1)
            return {"workload": "configs[1] sizes with Zipf(1.0) item popularity (SURVEY 8d's skewed distribution), "
                                "B = 2097152, SGD", "batch": rz["B"], "steps": rz["steps"],
                    "value": rz["steps"] * rz["B"] / rz["dt"], "unit": "interactions/s",
                    "ms_per_step": rz["dt"] / rz["steps"] * 1e3, "gpu_ms_per_step_events": gpuz,
                    "roofline": {"bound": "hbm", "achieved": achz, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": achz / HBM_PEAK_GBS, "traffic": None,
                                 "algorithmic_bytes_per_interaction": ALGO_BYTES_PER_INTERACTION_SGD(64)}}
        extra["mf_zipf"] = guarded("extra mf_zipf", zipf_leg, collective=False)
        extra["mf_ml100k_b256"] = guarded("extra mf_ml100k_b256", extra_small_batch, dev, collective=False)
        want_cpu_x = not a.no_cpu_baseline
        extra["neumf_ml1m"] = guarded("extra neumf_ml1m", extra_neumf, dev, want_cpu_x, collective=False)
        torch.cuda.empty_cache()
        extra["lightgcn_amazon_book"] = guarded("extra lightgcn_amazon_book", extra_lightgcn, dev, want_cpu_x,
                                                collective=False)
        torch.cuda.empty_cache()

    ref, ref_global = None, {}
    if world > 1 and wl in ("c3", "tiny") and not a.no_ref:
        # the same workload on ONE GPU (rank 0; the others wait), so that the N-GPU / 1-GPU ratio on BASELINE configs[2]
        # is on this line - at the single-GPU operating point (2M-interaction steps) AND at the N-GPU run's GLOBAL batch
        # (the loss is a batch sum and one GPU's throughput rises with B: only the second ratio compares like with like)
        if rank == 0:
            def one_gpu():
                d1 = build_data(a, 0, 1, dev, wl)

                def point(B1):
                    r1 = measure(a, d1, 0, 1, dev, B1, 1, None, a.warmup)
                    return {"value": r1["steps"] * r1["B"] / r1["dt"], "unit": "interactions/s", "n_gpus": 1,
                            "steps": r1["steps"], "ms_per_step": r1["dt"] / r1["steps"] * 1e3, "batch": r1["B"],
                            "workload": r1["name"], "batch_over_nnz": r1["B"] / r1["n"],
                            "roofline_frac": ALGO_BYTES_PER_INTERACTION_SGD(64) * r1["B"] / (r1["dt"] / r1["steps"]) / 1e9 / HBM_PEAK_GBS}
                base = point((1 << 21) if wl == "c3" else (1 << 16))
                at_global = {}
                locals_ = sorted({r["B"]} | {s_["B"] for s_ in sweep if "error" not in s_})
                for Bl in locals_:
                    Bg = min(Bl * world, d1["n"])            # (a global batch beyond the set: one step per epoch)
                    try:
                        at_global[Bl] = point(Bg)
                    except Exception as e:                  # noqa: BLE001  (the stage is B x 256 B: may not fit one GPU)
                        at_global[Bl] = {"error": f"{type(e).__name__}: {e}", "batch": Bg}
                        torch.cuda.empty_cache()
                free_data(d1)
                return base, at_global
            got = guarded("single-GPU reference", one_gpu, collective=False)   # (the other ranks wait at the barrier)
            if isinstance(got, dict):
                ref = got
            else:
                ref, ref_global = got
        dist.barrier()

    if watchdog is not None:
        watchdog.cancel()
    if rank == 0:
        B, d, steps, dt = r["B"], r["d"], r["steps"], r["dt"]
        value = steps * B * world / dt
        achieved, gpu_ms_mean = roofline_of(r, world)
        step_ms = r["step_ms"]
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath) and wl == "c2" and world == 1 and a.item_mode == "fused" and a.nnz is None:
            try:        # per-interaction figure of the committed PMC passes of THIS kernel chain x this run's batch
                t = json.load(open(tpath))
                traffic = t["hbm_bytes_per_interaction"] * B
                traffic_src = t.get("source", "profiles/pmc_traffic.json")
            except Exception:
                traffic = None
        if r["staged"]:
            chain = ("one SGD step = k_staged_user (forward + user rows) + its edge kernel (which also reduces the batch sums) + "
                     "k_staged_item (item rows in place; the next batch's pre-norm rides on it) + its edge kernel"
                     + (" + RCCL reduce-scatter / k_item_apply_counts / all-gather" if world > 1 else ""))
        else:
            chain = "one SGD step = k_fwd + k_reduce_partials + k_item_grad_" + a.item_mode + " + k_user + k_item_apply"
        out = {
            "metric": "BPR training interactions/sec at d=64; achieved HBM GB/s vs peak",
            "value": value, "unit": "interactions/s", "n_gpus": world, "steps": steps, "warmup": a.warmup,
            "ms_per_step": dt / steps * 1e3, "higher_is_better": True, "scaling": r["scaling"],
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": r["name"], "batch_per_gpu": B, "global_batch": B * world, "d": d,
                       "optimizer": "sgd", "lr": r["lr"], "reg_1": r["reg"], "reg_2": r["reg"], "loss": "BPR",
                       "item_mode": a.item_mode, "id_distribution": a.dist, "interactions_per_gpu": r["n"],
                       "parallelism": f"user-sharded dp{world}" if world > 1 else "single GPU",
                       "exchange_slices": r["slices"] if world > 1 else None,
                       "plan_layout": r["plan_kind"], "plan_bytes": r["plan_bytes"], "index_bytes": r["index_bytes"],
                       "plan_bytes_per_interaction": r["plan_bytes"] / r["n"], "plan_overlapped": bool(a.overlap_plan),
                       "semantics": "batch-synchronous (autograd + SGD.step equivalent), shuffle=True (device Feistel permutation per epoch)"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": chain + " (+ the epoch plan build amortised over its batches)",
                         "algorithmic_bytes_per_interaction": ALGO_BYTES_PER_INTERACTION_SGD(d),
                         "gpu_ms_per_step_events": gpu_ms_mean, "gpu_ms_per_step_median": step_ms[len(step_ms) // 2]},
        }
        if world == 1 and not a.no_extras:
            # what this box's HBM delivers to the simplest kernels there are, measured now (outside every timed region): the
            # runtime's device-to-device copy of 2 GiB (bytes read + written).  `peak` stays the guide's 8 TB/s; this says how
            # much of it a copy reaches here (profiles/r06_gather_probe.txt: 5.0 copy / 5.5 read-only / 5.3 write-only)
            try:
                nb = 1 << 31
                src = torch.empty(nb // 4, dtype=torch.float32, device=dev).fill_(1.0)
                dst = torch.empty_like(src)
                dst.copy_(src)
                _, ms_c = _timed(lambda: dst.copy_(src), 5)
                out["roofline"]["device_copy_GBs"] = 2.0 * nb * 5 / (ms_c * 1e-3) / 1e9
                del src, dst
            except Exception as e:          # informational only
                out["roofline"]["device_copy_GBs"] = None
                print(f"bench.py: device copy rate not measured: {e}", file=sys.stderr)
        if len(r["all_dt"]) > 1:          # every timed region of `steps` steps; `value` / `ms_per_step` are the median one's
            out["repeats"] = [steps * B * world / t_ for t_ in r["all_dt"]]
            out["repeats_note"] = (f"{len(r['all_dt'])} timed regions of {steps} steps each, every one bracketed by barrier + "
                                   "synchronize; value = the median region")
        if extra:
            out["extra"] = extra
        if secondary is not None:
            out["secondary"] = secondary
        if main_note is not None:
            out["pipelined_exchange_failed"] = main_note
        if world > 1:
            out["distributed"] = diag
            out["item_exchange"] = {"main_point_wire_bytes_per_step_and_rank": r.get("wire_bytes"), "small_batch_points": xchg}
            if r["split_ms"] is not None:
                out["step_split_ms"] = {"compute": r["split_ms"][0], "exposed_exchange": r["split_ms"][1],
                                        "note": "HIP events on the step's stream around the item exchange (rank 0): "
                                                "compute = the step's kernels + the two small all-reduces they wait for; "
                                                "exposed exchange = reduce-scatter + owner update + all-gather as far as "
                                                "they did not hide under the item pass (slices > 1)"}
            if check is not None:
                out["replica_check"] = check
            out["six_x_budget"] = ("DESIGN.md section 5: the exchange is 2 x (N-1)/N x I x 264 B per step and rank whatever the "
                                   "batch (replicated Q), so >= 6x at 8 GPUs is budgeted only for B_local >= 8M interactions "
                                   "per rank and step (about 3x at 2M); the sweep measures it")
        if world > 1:
            out["config"]["global_batch_over_nnz"] = B * world / (r["n"] * world)
        if ref is not None:
            out["single_gpu_same_workload"] = ref
            if "error" in ref:
                ref = None
            else:
                out["speedup_vs_single_gpu_same_workload"] = value / ref["value"]
                out["speedup_note"] = ("speedup_vs_single_gpu_same_workload divides by the 1-GPU rate at ITS operating point "
                                       f"(B = {ref['batch']}); speedup_vs_1gpu_same_global_batch divides by the 1-GPU rate at "
                                       "this run's GLOBAL batch (same optimisation trajectory: the loss is a batch sum)")
            g = ref_global.get(B)
            if g is not None:
                out["single_gpu_same_global_batch"] = g
                if "error" not in g:
                    out["speedup_vs_1gpu_same_global_batch"] = value / g["value"]
        if sweep:
            pts = []
            for s_ in sweep:
                if "error" in s_:
                    pts.append(s_)
                    continue
                v = s_["steps"] * s_["B"] * world / s_["dt"]
                pt = {"batch_per_gpu": s_["B"], "slices": s_["slices"], "steps": s_["steps"], "value": v,
                      "ms_per_step": s_["dt"] / s_["steps"] * 1e3,
                      "compute_ms": s_["split_ms"][0] if s_["split_ms"] else None,
                      "exposed_exchange_ms": s_["split_ms"][1] if s_["split_ms"] else None}
                pt["global_batch"] = s_["B"] * world
                pt["global_batch_over_nnz"] = s_["B"] / s_["n"]
                if ref is not None:
                    pt["speedup_vs_1gpu"] = v / ref["value"]
                    pt["meets_6x"] = bool(v / ref["value"] >= 6.0)
                g = ref_global.get(s_["B"])
                if g is not None and "error" not in g:
                    pt["one_gpu_value_same_global_batch"] = g["value"]
                    pt["speedup_vs_1gpu_same_global_batch"] = v / g["value"]
                    pt["meets_6x_same_global_batch"] = bool(v / g["value"] >= 6.0)
                pts.append(pt)
            out["sweep"] = pts
        if r.get("cpu_result") is not None:
            out["cpu_baseline"] = r["cpu_result"]
        elif r["cpu_batches"]:
            out["cpu_baseline"] = cpu_baseline(r["U"], r["I"], d, B, r["cpu_batches"], r["reg"])
        emit(out)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _spawned(local_rank, world, port, argv):
    os.environ.update(RANK=str(local_rank), LOCAL_RANK=str(local_rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.argv = [sys.argv[0]] + list(argv)
    main()


if __name__ == "__main__":
    main()
