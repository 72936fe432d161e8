"""The user-sharded multi-GPU protocol with the REAL kernels: two ranks share the one GPU of the test box,
collectives over gloo (RCCL refuses two ranks on one device; the RCCL path itself is covered at world size 1
in test_gpu_parity.py).  Each rank owns a user range and its P rows; together they must reproduce the
single-process oracle step on the union batch, through both entry points of the trainer (triples / plan)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import bpr_mf_numpy as O

pytestmark = pytest.mark.gpu
U, I, D, B, STEPS = 400, 300, 64, 4096, 3
LR, R1, R2 = 0.05, 0.01, 0.02


def _data(items=None):
    rng = np.random.default_rng(11)
    I_ = items or I
    P0 = (rng.standard_normal((U, D)) * 0.2).astype(np.float32)
    Q0 = (rng.standard_normal((I_, D)) * 0.2).astype(np.float32)
    batches = [np.stack([rng.integers(0, U, B), rng.integers(0, I_, B), rng.integers(0, I_, B)], 1).astype(np.int32)
               for _ in range(STEPS)]
    return P0, Q0, batches


def _worker(rank, world, port, out_dir, use_plan, mode, backend="gloo", slices=1, exchange="dense", items=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    from daisyrec_amd import ops
    from daisyrec_amd.sharding import UserShardedBprTrainer, shard_triples, user_range
    dev = torch.device("cuda", rank if backend == "nccl" else 0)
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    P0, Q0, batches = _data(items)
    I = Q0.shape[0]
    lo, hi = user_range(U, world, rank)
    P = torch.from_numpy(P0[lo:hi].copy()).to(dev)
    Q = torch.from_numpy(Q0.copy()).to(dev)
    ctx = ops.BprContext(B, D, hi - lo, I, device=dev)
    tr = UserShardedBprTrainer(ctx, P, Q, lo, LR, R1, R2, item_mode=ops.ITEM_MODES[mode], slices=slices, exchange=exchange,
                               global_batch=B)
    assert tr.slices == (slices if mode == "fused" and not tr.sparse else 1)
    assert tr.staged == (mode == "fused") and tr.sparse == (exchange == "sparse")
    losses = []
    for b in batches:
        mine = torch.from_numpy(shard_triples(b, U, world, rank)).to(dev)
        if use_plan:                      # one-batch plan of the rank's share (local user ids)
            loc = mine.clone()
            loc[:, 0] -= lo
            plan = ops.EpochPlan(loc.shape[0], hi - lo, I, device=dev)
            index = None
            if mode == "fused":           # the partitioned layout the staged step normally reads
                index = ops.TrainIndex(loc, hi - lo, I)
                plan.build_indexed(index, loc.shape[0], order="identity")
            else:
                plan.build(loc, loc.shape[0], order="identity")
            stats = tr.step_from_plan(plan, 0)
            torch.cuda.synchronize()
            plan.close()
            if index is not None:
                index.close()
        else:
            stats = tr.step_from_triples(mine)
        losses.append(float(stats[7].cpu()))
    torch.cuda.synchronize()
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), P=P.cpu().numpy(), Q=Q.cpu().numpy(), lo=lo, hi=hi,
             losses=np.array(losses))
    ctx.close()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _check(tmp_path, world, items=None):
    P, Q, batches = _data(items)
    ref = []
    for b in batches:
        loss, P, Q = O.mf_sgd_step(P, Q, b[:, 0], b[:, 1], b[:, 2], LR, R1, R2)
        ref.append(loss)
    outs = [np.load(os.path.join(str(tmp_path), f"r{r}.npz")) for r in range(world)]
    for o in outs:
        np.testing.assert_allclose(o["losses"], ref, rtol=1e-6)                  # every rank sees the GLOBAL loss
        np.testing.assert_allclose(o["Q"], Q, atol=1e-5)
        np.testing.assert_allclose(o["P"], P[int(o["lo"]):int(o["hi"])], atol=1e-5)
    for o in outs[1:]:
        np.testing.assert_allclose(outs[0]["Q"], o["Q"], atol=1e-6)              # replicas stay together


@pytest.mark.parametrize("mode", ["fused", "chunked"])
@pytest.mark.parametrize("use_plan", [False, True])
def test_two_ranks_on_one_gpu_equal_the_single_process_step(tmp_path, use_plan, mode):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), use_plan, mode), nprocs=world, join=True)
    _check(tmp_path, world)


@pytest.mark.parametrize("world,slices", [(2, 4), (3, 5)])
def test_item_pass_in_slices_with_the_exchange_on_a_side_stream(tmp_path, world, slices):
    """slices > 1: the item pass range by range, each range's reduce-scatter / owner update / all-gather queued on a
    side stream behind it; ownership interleaves per range; 300 items are not divisible by 3 x 5 (padded staging)"""
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), True, "fused", "gloo", slices), nprocs=world, join=True)
    _check(tmp_path, world)


@pytest.mark.parametrize("world,use_plan", [(2, True), (3, False)])
def test_touched_rows_exchange_with_the_real_kernels(tmp_path, world, use_plan):
    """exchange='sparse' (round 6): 2 B = 8192 of 100 003 item rows per step - id lists all-gathered, the union merged on
    every rank, reduce-scatter / k_item_apply_counts on the union's owner blocks / all-gather; the staged item pass writes
    gQ / cnt with the zero row behind them"""
    items = 100003
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), use_plan, "fused", "gloo", 1, "sparse", items), nprocs=world,
             join=True)
    _check(tmp_path, world, items)


def test_item_rows_not_divisible_by_the_world_size(tmp_path):
    """the staged protocol's padded path: a world size that does not divide the item count"""
    world = 4 if I % 4 else 7
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), False, "fused"), nprocs=world, join=True)
    _check(tmp_path, world)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs one GPU per rank (this box has one)")
@pytest.mark.parametrize("mode,exchange", [("fused", "dense"), ("chunked", "dense"), ("fused", "sparse")])
def test_rccl_ranks_equal_the_single_process_step(tmp_path, mode, exchange):
    """The same check over the real collectives (reduce_scatter_tensor / all_gather_into_tensor / all_reduce
    on RCCL), one rank per GPU: runs on the first box that has more than one GPU."""
    world = min(torch.cuda.device_count(), 8)
    items = 100003 if exchange == "sparse" else None
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), True, mode, "nccl", 1, exchange, items), nprocs=world, join=True)
    _check(tmp_path, world, items)


def test_bench_refuses_more_gpus_than_the_box_has():
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    want = torch.cuda.device_count() + 1
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(want), "--workload", "tiny"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "refusing" in (r.stderr + r.stdout)
    assert '"n_gpus"' not in r.stdout


def test_bench_self_spawns_its_ranks():
    """`python bench.py --gpus 2` without a launcher spawns 2 ranks (gloo here: both share the one GPU)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--workload", "tiny",
                        "--backend", "gloo", "--batch", "65536", "--steps", "6", "--warmup", "2"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 2 * 65536 and out["value"] > 0
    # the self-diagnosing part of a multi-rank run: who took part, are the replicas identical and
    # equal to the single-GPU fit, where does a step's time go, the B_local x slices sweep with its speed-ups
    dd = out["distributed"]
    assert dd["world_size"] == 2 and len(dd["ranks"]) == 2 and dd["backend"] == "gloo" and dd["rccl_version"]
    rc = out["replica_check"]
    assert rc["q_checksum_identical_on_all_ranks"] is True and rc["loss_equal_to_1e-6"] is True, rc
    # (an element crossing zero flips the L1 regulariser's sign a step apart in the two runs: rare jumps of ~1e-4)
    assert rc["q_max_abs_diff_vs_1gpu"] < 1e-3 and rc["p_max_abs_diff_vs_1gpu"] < 1e-3, rc
    assert rc["q_fraction_beyond_1e-5"] < 1e-3 and rc["p_fraction_beyond_1e-5"] < 1e-3, rc
    sp = out["step_split_ms"]
    assert sp["compute"] > 0 and sp["exposed_exchange"] >= 0
    assert out["single_gpu_same_workload"]["value"] > 0 and out["speedup_vs_single_gpu_same_workload"] > 0
    pts = out["sweep"]
    assert {(p["batch_per_gpu"], p["slices"]) for p in pts} == {(16384, 1), (16384, 4), (65536, 1), (65536, 4)}
    assert all(p["value"] > 0 and p["compute_ms"] > 0 and "meets_6x" in p for p in pts)
    # like with like: the 1-GPU rate at the run's GLOBAL batch next to the one at its own operating
    # point - on the line and on every sweep point
    g = out["single_gpu_same_global_batch"]
    assert g["batch"] == 2 * 65536 and g["value"] > 0 and out["speedup_vs_1gpu_same_global_batch"] > 0
    assert 0 < out["config"]["global_batch_over_nnz"] <= 1
    assert all(p["global_batch"] == 2 * p["batch_per_gpu"] and p["speedup_vs_1gpu_same_global_batch"] > 0
               and "meets_6x_same_global_batch" in p for p in pts)
