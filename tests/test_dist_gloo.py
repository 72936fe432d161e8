"""world_size-2 `gloo` test of the user-sharded protocol (daisyrec_amd/sharding.py) on CPU:
two ranks, each with its user range and the oracle-backed step backend, must reproduce the
single-process oracle step on the UNION batch (tables and global loss)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import bpr_mf_numpy as O

U, I, D, B, STEPS = 40, 30, 16, 96, 3
LR, R1, R2 = 0.05, 0.01, 0.02


def _data(lopsided=False):
    rng = np.random.default_rng(9)
    P0 = (rng.standard_normal((U, D)) * 0.2).astype(np.float32)
    Q0 = (rng.standard_normal((I, D)) * 0.2).astype(np.float32)
    batches = [np.stack([rng.integers(0, U, B), rng.integers(0, I, B), rng.integers(0, I, B)], 1)
               .astype(np.int32) for _ in range(STEPS)]
    if lopsided:                              # step 1: every sample belongs to the first rank's users
        batches[1][:, 0] = rng.integers(0, U // 4, B)
    return P0, Q0, batches


def _worker(rank, world, port, out_dir, mode, lopsided=False, slices=1, adam=False, exchange="dense"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from daisyrec_amd.sharding import UserShardedBprTrainer, shard_triples, user_range
    from oracle_backend import OracleContext
    P0, Q0, batches = _data(lopsided)
    lo, hi = user_range(U, world, rank)
    P = torch.from_numpy(P0[lo:hi].copy())
    Q = torch.from_numpy(Q0.copy())
    from daisyrec_amd import _native as N
    ctx = OracleContext(B, D, hi - lo, I)
    tr = UserShardedBprTrainer(ctx, P, Q, lo, LR, R1, R2, overlap=(rank % 2 == 0),
                               item_mode={"fused": N.ITEM_FUSED, "chunked": N.ITEM_CHUNKED}[mode], slices=slices,
                               adam_steps=2 if adam else 0,       # (the table of step constants grows on demand)
                               exchange=exchange, global_batch=B)
    assert tr.staged == (mode == "fused") and tr.slices == (slices if mode == "fused" and not tr.sparse else 1)
    assert tr.sparse == (exchange == "sparse")
    losses = []
    for b in batches:
        mine = shard_triples(b, U, world, rank)
        if len(mine) == 0:                    # this rank owns no sample of the step: it only joins the exchanges
            stats = tr.step_from_plan(None, 0)
        else:
            stats = tr.step_from_triples(torch.from_numpy(mine))
        losses.append(float(stats[7]))
    if adam:
        ctx.oracle_flush_p(P, tr.adam)         # (the HIP path: ShardedAdam.flush - the rows the last steps did not reference)
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), P=P.numpy(), Q=Q.numpy(), lo=lo, hi=hi,
             losses=np.array(losses), acc=float(ctx.epoch_acc[0]))
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


import pytest


@pytest.mark.parametrize("mode,world", [("fused", 2), ("chunked", 2), ("fused", 4)])
def test_user_sharding_equals_single_process(tmp_path, mode, world):
    """staged protocol (reduce-scatter / owner apply / all-gather; world 3: item rows not divisible by the
    world size) and the phase protocol (dense all-reduce)"""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), mode), nprocs=world, join=True)
    P, Q, batches = _data()
    ref_losses = []
    for b in batches:
        loss, P, Q = O.mf_sgd_step(P, Q, b[:, 0], b[:, 1], b[:, 2], LR, R1, R2)
        ref_losses.append(loss)
    outs = [np.load(os.path.join(str(tmp_path), f"r{r}.npz")) for r in range(world)]
    for o in outs:
        np.testing.assert_allclose(o["losses"], ref_losses, rtol=1e-9)       # every rank sees the GLOBAL loss
        np.testing.assert_allclose(o["Q"], Q, atol=2e-6)                     # replicas stay identical
        np.testing.assert_allclose(o["P"], P[int(o["lo"]):int(o["hi"])], atol=2e-6)
        assert abs(float(o["acc"]) - sum(ref_losses)) < 1e-6
    for o in outs[1:]:
        np.testing.assert_array_equal(outs[0]["Q"], o["Q"])


def test_ranks_without_samples_in_a_step_only_join_the_exchanges(tmp_path):
    """a rank's share of a global batch can be empty (MF.fit over ranks: EpochPlan.build_positions)"""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    world = 4
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), "fused", True), nprocs=world, join=True)
    P, Q, batches = _data(True)
    ref_losses = []
    for b in batches:
        loss, P, Q = O.mf_sgd_step(P, Q, b[:, 0], b[:, 1], b[:, 2], LR, R1, R2)
        ref_losses.append(loss)
    for r in range(world):
        o = np.load(os.path.join(str(tmp_path), f"r{r}.npz"))
        np.testing.assert_allclose(o["losses"], ref_losses, rtol=1e-9)
        np.testing.assert_allclose(o["Q"], Q, atol=2e-6)
        np.testing.assert_allclose(o["P"], P[int(o["lo"]):int(o["hi"])], atol=2e-6)


# ---- the touched-rows item exchange (sharding.py: exchange='sparse'; SURVEY 8e, north_star "only where users overlap items")
SP_U, SP_I, SP_B = 60, 500, 12            # 2 B << I: at most 24 of 500 item rows move per step


def _sparse_data(kind):
    """kind 'few': 2 B << I, step 1 leaves every rank but the first without samples and step 2 repeats one item in
    every sample (a union of 1 + B rows, hit by every rank); 'all': the union is the whole table (I = 30 <= 2 B)"""
    if kind == "all":
        return (U, I, B) + _data(True)
    rng = np.random.default_rng(17)
    P0 = (rng.standard_normal((SP_U, D)) * 0.2).astype(np.float32)
    Q0 = (rng.standard_normal((SP_I, D)) * 0.2).astype(np.float32)
    batches = [np.stack([rng.integers(0, SP_U, SP_B), rng.integers(0, SP_I, SP_B), rng.integers(0, SP_I, SP_B)], 1)
               .astype(np.int32) for _ in range(4)]
    batches[1][:, 0] = rng.integers(0, SP_U // 8, SP_B)
    batches[2][:, 1] = 7
    batches[3] = batches[3][:5]                         # the epoch's partial last batch
    return SP_U, SP_I, SP_B, P0, Q0, batches


def _sparse_worker(rank, world, port, out_dir, kind, exchange):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from daisyrec_amd.sharding import UserShardedBprTrainer, shard_triples, user_range
    from daisyrec_amd import _native as N
    from oracle_backend import OracleContext
    U_, I_, B_, P0, Q0, batches = _sparse_data(kind)
    lo, hi = user_range(U_, world, rank)
    P, Q = torch.from_numpy(P0[lo:hi].copy()), torch.from_numpy(Q0.copy())
    ctx = OracleContext(B_, D, hi - lo, I_)             # (a rank may hold the whole global batch of a step)
    tr = UserShardedBprTrainer(ctx, P, Q, lo, LR, R1, R2, item_mode=N.ITEM_FUSED, overlap=(rank % 2 == 0), slices=3,
                               exchange=exchange, global_batch=B_)
    losses, moved = [], []
    for b in batches:
        mine = shard_triples(b, U_, world, rank)
        q_before = Q.clone()
        stats = tr.step_from_plan(None, 0) if len(mine) == 0 else tr.step_from_triples(torch.from_numpy(mine))
        losses.append(float(stats[7]))
        moved.append(int((Q != q_before).any(1).sum()))
    assert float(tr.gQ.abs().sum()) == 0.0 and float(tr.cnt.abs().sum()) == 0.0     # re-zeroed behind every exchange
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), P=P.numpy(), Q=Q.numpy(), lo=lo, hi=hi, losses=np.array(losses),
             moved=np.array(moved), sparse=int(tr.sparse), slices=tr.slices, wire=tr.wire_bytes[tr.wire_bytes["used"]],
             caps=np.array([tr.cap_local, tr.cap_union, tr.rows_s]) if tr.sparse else np.zeros(3))
    dist.destroy_process_group()


@pytest.mark.parametrize("kind,world,exchange", [("few", 2, "sparse"), ("few", 4, "sparse"), ("all", 4, "sparse"),
                                                 ("few", 3, "auto"), ("all", 2, "auto")])
def test_touched_rows_exchange_equals_single_process(tmp_path, kind, world, exchange):
    """only the union of the ranks' touched item rows travels (index lists all-gathered, the union merged identically on
    every rank, reduce-scatter / owner apply / all-gather over |union| rows): against the oracle step on the union batch,
    with a rank that holds no sample of a step, items only one rank touches, an item every rank touches, and a union that
    is the whole table; 'auto' picks sparse for 2 B << I and dense when the table is smaller than the batch"""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    mp.spawn(_sparse_worker, args=(world, _free_port(), str(tmp_path), kind, exchange), nprocs=world, join=True)
    U_, I_, B_, P, Q, batches = _sparse_data(kind)
    ref_losses, touched = [], []
    for b in batches:
        loss, P, Q = O.mf_sgd_step(P, Q, b[:, 0], b[:, 1], b[:, 2], LR, R1, R2)
        ref_losses.append(loss)
        touched.append(len(set(b[:, 1]) | set(b[:, 2])))
    outs = [np.load(os.path.join(str(tmp_path), f"r{r}.npz")) for r in range(world)]
    want_sparse = exchange == "sparse" or kind == "few"
    for o in outs:
        assert bool(o["sparse"]) == want_sparse and int(o["slices"]) == (1 if want_sparse else 3)
        np.testing.assert_allclose(o["losses"], ref_losses, rtol=1e-9)
        np.testing.assert_allclose(o["Q"], Q, atol=2e-6)
        np.testing.assert_allclose(o["P"], P[int(o["lo"]):int(o["hi"])], atol=2e-6)
        assert list(o["moved"]) == touched              # exactly the touched rows of Q changed in every step
        if want_sparse:
            cap_local, cap_union, rows = (int(x) for x in o["caps"])
            assert cap_local == min(I_, 2 * B_) and cap_union == rows * world and cap_union >= min(I_, 2 * B_)
    for o in outs[1:]:
        np.testing.assert_array_equal(outs[0]["Q"], o["Q"])                  # replicas stay bit-identical
    if kind == "few":                                   # the point of it: a fraction of the dense exchange's bytes
        from daisyrec_amd.sharding import item_exchange_bytes
        b = item_exchange_bytes(I_, D, world, B_, B_)
        assert int(outs[0]["wire"]) == b["sparse"] and b["sparse"] * 5 < b["dense"]


def _disagree_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from daisyrec_amd.sharding import UserShardedBprTrainer, user_range
    from daisyrec_amd import _native as N
    from oracle_backend import OracleContext
    lo, hi = user_range(U, world, rank)
    P, Q = torch.zeros(hi - lo, D), torch.zeros(I, D)
    try:
        UserShardedBprTrainer(OracleContext(B, D, hi - lo, I), P, Q, lo, LR, R1, R2, item_mode=N.ITEM_FUSED, slices=2 + rank)
        msg = "no error"
    except RuntimeError as e:
        msg = str(e)
    open(os.path.join(out_dir, f"r{rank}.txt"), "w").write(msg)
    dist.destroy_process_group()


def test_ranks_that_disagree_on_the_slices_are_told_so(tmp_path):
    """a slice count resolved per rank (environment, arguments) is compared across the ranks instead of trusted"""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    mp.spawn(_disagree_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        assert "disagree on the exchange" in open(os.path.join(str(tmp_path), f"r{r}.txt")).read()


@pytest.mark.parametrize("world,slices", [(2, 3), (4, 5)])
def test_sliced_exchange_equals_single_process(tmp_path, world, slices):
    """slices > 1 (sharding.py: item pass range by range, each range's exchange right behind it, rank r owning the
    r-th block of every range): 30 items over world x slices blocks needs padding in both cases; one step has
    ranks without samples"""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), "fused", True, slices), nprocs=world, join=True)
    P, Q, batches = _data(True)
    ref_losses = []
    for b in batches:
        loss, P, Q = O.mf_sgd_step(P, Q, b[:, 0], b[:, 1], b[:, 2], LR, R1, R2)
        ref_losses.append(loss)
    outs = [np.load(os.path.join(str(tmp_path), f"r{r}.npz")) for r in range(world)]
    for o in outs:
        np.testing.assert_allclose(o["losses"], ref_losses, rtol=1e-9)
        np.testing.assert_allclose(o["Q"], Q, atol=2e-6)
        np.testing.assert_allclose(o["P"], P[int(o["lo"]):int(o["hi"])], atol=2e-6)
    for o in outs[1:]:
        np.testing.assert_array_equal(outs[0]["Q"], o["Q"])


@pytest.mark.parametrize("world,slices", [(2, 1), (4, 3)])
def test_sharded_adam_equals_single_process_dense_adam(tmp_path, world, slices):
    """torch.optim.Adam through the sharded protocol (trainer orchestration only; the kernels are tested on the GPU):
    per step the catch-up / user pass with Adam on every rank's rows of P, the gradient form of the item pass, the
    reduce-scatter, the owner's DENSE Adam on its block of Q - with slices > 1 a rank owns one block per slice, each with
    its own moments - and the all-gather; a step in which some ranks own no sample; against oracle.DenseAdam on the
    union batches."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), "fused", True, slices, True), nprocs=world, join=True)
    P, Q, batches = _data(True)
    opt = O.DenseAdam([P.shape, Q.shape], LR)
    ref_losses = []
    for b in batches:
        loss, gP, gQ = O.mf_pair_grad(P, Q, b[:, 0], b[:, 1], b[:, 2], R1, R2)
        P, Q = opt.step([P, Q], [gP, gQ])
        ref_losses.append(loss)
    outs = [np.load(os.path.join(str(tmp_path), f"r{r}.npz")) for r in range(world)]
    for o in outs:
        np.testing.assert_allclose(o["losses"], ref_losses, rtol=1e-7)
        np.testing.assert_allclose(o["Q"], Q, atol=1e-5)
        np.testing.assert_allclose(o["P"], P[int(o["lo"]):int(o["hi"])], atol=1e-5)
    for o in outs[1:]:
        np.testing.assert_array_equal(outs[0]["Q"], o["Q"])


def test_user_range_partition():
    from daisyrec_amd.sharding import shard_triples, user_range
    for U_, W in ((10, 3), (8, 8), (1000003, 8), (5, 8)):
        r = [user_range(U_, W, k) for k in range(W)]
        assert r[0][0] == 0 and r[-1][1] == U_
        assert all(a[1] == b[0] for a, b in zip(r, r[1:]))
        assert max(h - l for l, h in r) - min(h - l for l, h in r) <= 1
    t = np.array([[0, 1, 2], [9, 1, 2], [4, 0, 0]], dtype=np.int32)
    assert sum(len(shard_triples(t, 10, 3, k)) for k in range(3)) == 3


# ---- LightGCN: the row-sharded product (sharding.RowShardedPropagation.spmm) on CPU ---------------------------
LU, LI, LD, LNNZ = 37, 26, 8, 400          # 63 nodes: divisible neither by 2 nor by 3 ranks x 3 pieces


def _lg_data():
    rng = np.random.default_rng(21)
    gu, gi = rng.integers(0, LU, LNNZ), rng.integers(0, LI, LNNZ)
    gu[gu == 5] = 6                         # isolated nodes: user 5 and item 0 have no edge
    gi[gi == 0] = 1
    X = rng.standard_normal((LU + LI, LD)).astype(np.float32)
    return gu, gi, X


def _lg_worker(rank, world, port, out_dir, pieces):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from daisyrec_amd.sharding import RowShardedPropagation
    from oracle_backend import OracleGraph
    gu, gi, X = _lg_data()
    N = LU + LI
    prop = RowShardedPropagation(OracleGraph(gu, gi, LU, LI), N, LD, "cpu", pieces=pieces)
    assert prop.pieces == pieces and prop.rows * world >= N and prop.side is None
    x = torch.from_numpy(X)
    y1 = prop.spmm(x, torch.empty(N, LD)).clone()
    y2 = prop.spmm(y1, torch.empty(N, LD)).clone()          # a second layer through the same buffers
    np.savez(os.path.join(out_dir, f"lg{rank}.npz"), y1=y1.numpy(), y2=y2.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("world,pieces", [(2, 1), (2, 3), (3, 1), (3, 3)])
def test_row_sharded_lightgcn_product_on_cpu(tmp_path, world, pieces):
    """every rank reduces its node range (in `pieces` sub-blocks) and all-gathers the blocks: all ranks end with the
    whole product in node order, padded tails and isolated nodes included (LightGCNRecommender.py:117-129)"""
    from oracle import lightgcn_numpy as LG
    mp.spawn(_lg_worker, args=(world, _free_port(), str(tmp_path), pieces), nprocs=world, join=True)
    gu, gi, X = _lg_data()
    csr = LG.norm_adj_csr(gu, gi, LU, LI)
    want1 = LG.spmm(csr, X.astype(np.float64))
    want2 = LG.spmm(csr, want1)
    for r in range(world):
        o = np.load(os.path.join(str(tmp_path), f"lg{r}.npz"))
        np.testing.assert_allclose(o["y1"], want1, atol=1e-6)
        np.testing.assert_allclose(o["y2"], want2, atol=1e-5)


def _fm_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from daisyrec_amd.sharding import UserShardedBprTrainer, shard_triples, user_range
    from daisyrec_amd import _native as N
    from oracle_backend import OracleContext
    P0, Q0, batches = _data(True)
    bu0, bi0, b00 = _fm_biases()
    lo, hi = user_range(U, world, rank)
    P, Q = torch.from_numpy(P0[lo:hi].copy()), torch.from_numpy(Q0.copy())
    bu, bi, b0 = torch.from_numpy(bu0[lo:hi].copy()), torch.from_numpy(bi0.copy()), torch.from_numpy(b00.copy())
    ctx = OracleContext(B, D, hi - lo, I)
    ctx.set_bias(bu, bi, b0, g_i_bias=torch.zeros(I))
    tr = UserShardedBprTrainer(ctx, P, Q, lo, LR, R1, R2, item_mode=N.ITEM_FUSED, slices=1 + rank % 1)
    assert tr.staged and tr.fm is not None
    losses = []
    for b in batches:
        mine = shard_triples(b, U, world, rank)
        stats = tr.step_from_plan(None, 0) if len(mine) == 0 else tr.step_from_triples(torch.from_numpy(mine))
        losses.append(float(stats[7]))
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), P=P.numpy(), Q=Q.numpy(), bu=bu.numpy(), bi=bi.numpy(), b0=b0.numpy(),
             lo=lo, hi=hi, losses=np.array(losses))
    dist.destroy_process_group()


def _fm_biases():
    rng = np.random.default_rng(21)
    return ((rng.standard_normal(U) * 0.1).astype(np.float32), (rng.standard_normal(I) * 0.1).astype(np.float32),
            np.array([0.05], np.float32))


@pytest.mark.parametrize("world", [2, 4])
def test_fm_user_sharding_equals_single_process(tmp_path, world):
    """FM's three bias parameters through the sharded protocol: u_bias rows with their users, the
    item-bias gradient all-reduced, bias_ from the all-reduced coefficient sum; one step leaves ranks without samples
    (they still join the bias exchanges) - against oracle.fm_numpy.fm_sgd_step on the union batches"""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from oracle import fm_numpy as F
    mp.spawn(_fm_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    P, Q, batches = _data(True)
    cur = [P, Q, *_fm_biases()]
    ref_losses = []
    for b in batches:
        loss, *cur = F.fm_sgd_step(*cur, b[:, 0], b[:, 1], b[:, 2], LR, R1, R2)
        ref_losses.append(loss)
    P, Q, bu, bi, b0 = cur
    outs = [np.load(os.path.join(str(tmp_path), f"r{r}.npz")) for r in range(world)]
    for o in outs:
        lo, hi = int(o["lo"]), int(o["hi"])
        np.testing.assert_allclose(o["losses"], ref_losses, rtol=1e-6)
        np.testing.assert_allclose(o["Q"], Q, atol=3e-6)
        np.testing.assert_allclose(o["P"], P[lo:hi], atol=3e-6)
        np.testing.assert_allclose(o["bu"], np.asarray(bu).reshape(-1)[lo:hi], atol=3e-6)
        np.testing.assert_allclose(o["bi"], np.asarray(bi).reshape(-1), atol=3e-6)
        np.testing.assert_allclose(o["b0"], np.asarray(b0).reshape(-1), atol=3e-6)
    for o in outs[1:]:
        np.testing.assert_array_equal(outs[0]["bi"], o["bi"])
        np.testing.assert_array_equal(outs[0]["b0"], o["b0"])


def _dense_worker(rank, world, port, out_dir, kind):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from daisyrec_amd.sharding import UserShardedBprTrainer, shard_triples, user_range
    from daisyrec_amd import _native as N
    from oracle_backend import OracleContext, OracleDense
    P0, Q0, batches = _data(True)
    lo, hi = user_range(U, world, rank)
    P, Q = torch.from_numpy(P0[lo:hi].copy()), torch.from_numpy(Q0.copy())
    ctx = OracleContext(B, D, hi - lo, I)
    tr = UserShardedBprTrainer(ctx, P, Q, lo, 0.01, R1, R2, item_mode=N.ITEM_FUSED, overlap=(rank % 2 == 0),
                               dense_opt=OracleDense(kind, 0.01))
    assert not tr.staged and tr.dense is not None
    losses = []
    for b in batches:
        mine = shard_triples(b, U, world, rank)
        stats = tr.step_from_plan(None, 0) if len(mine) == 0 else tr.step_from_triples(torch.from_numpy(mine))
        losses.append(float(stats[7]))
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), P=P.numpy(), Q=Q.numpy(), lo=lo, hi=hi, losses=np.array(losses))
    dist.destroy_process_group()


@pytest.mark.parametrize("kind,world", [("adagrad", 2), ("rmsprop", 4), ("adam", 3)])
def test_dense_optimiser_protocol_equals_single_process(tmp_path, kind, world):
    """the dense-optimiser protocol of the sharded trainer (torch's Adagrad / RMSprop / dense Adam behind the phase kernels:
    all-reduce of the dense item gradient, the optimiser on the rank's rows of P and on the replicated Q; a rank without a
    sample still steps its rows - RMSprop's and Adam's state moves without a gradient) against the oracle's dense
    optimisers on the union batches"""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    mp.spawn(_dense_worker, args=(world, _free_port(), str(tmp_path), kind), nprocs=world, join=True)
    P, Q, batches = _data(True)
    opt = {"adam": O.DenseAdam, "adagrad": O.DenseAdagrad, "rmsprop": O.DenseRMSprop}[kind]([P.shape, Q.shape], 0.01)
    ref_losses = []
    for b in batches:
        loss, gP, gQ = O.mf_pair_grad(P, Q, b[:, 0], b[:, 1], b[:, 2], R1, R2)
        P, Q = opt.step([P, Q], [gP, gQ])
        ref_losses.append(loss)
    outs = [np.load(os.path.join(str(tmp_path), f"r{r}.npz")) for r in range(world)]
    for o in outs:
        np.testing.assert_allclose(o["losses"], ref_losses, rtol=1e-6)
        np.testing.assert_allclose(o["Q"], Q, atol=2e-5)
        np.testing.assert_allclose(o["P"], P[int(o["lo"]):int(o["hi"])], atol=2e-5)
    for o in outs[1:]:
        np.testing.assert_array_equal(outs[0]["Q"], o["Q"])



def _auto_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from daisyrec_amd.sharding import UserShardedBprTrainer, auto_exchange_slices, user_range
    from oracle_backend import OracleContext
    # the rule itself, priced as if the job ran over RCCL, from numbers every rank holds - all-gathered and compared
    cases = [(1_000_000, 64, 8, 1 << 21), (1_000_000, 64, 8, 1 << 24), (100_000, 64, 8, 1 << 21), (100_000, 64, 2, 1 << 16),
             (3706, 64, 8, 65536), (1_000_000, 128, 4, 1 << 22), (5000, 16, world, 96)]
    mine = [auto_exchange_slices(*c, backend="nccl") for c in cases]
    everyone = [None] * world
    dist.all_gather_object(everyone, mine)
    # and through the trainer on this (gloo) job: 'auto' must resolve to ONE slice here and identically on every rank,
    # whatever a rank's own share of the batch is (the context of rank 1 is larger on purpose)
    lo, hi = user_range(U, world, rank)
    ctx = OracleContext(B * (1 + rank), D, hi - lo, I)
    P = torch.zeros(hi - lo, D)
    Q = torch.zeros(I, D)
    tr = UserShardedBprTrainer(ctx, P, Q, lo, LR, R1, R2, slices="auto", auto_batch=B // world)
    got = [None] * world
    dist.all_gather_object(got, tr.slices)
    if rank == 0:
        np.savez(os.path.join(out_dir, "auto.npz"), rule=np.array(everyone), trainer=np.array(got))
    dist.destroy_process_group()


def test_automatic_exchange_slices_are_the_same_on_every_rank(tmp_path):
    """`slices='auto'` (sharding.auto_exchange_slices) is a function of the item table, the world size
    and the per-rank share of the GLOBAL batch only - every rank of a 2-rank gloo job computes the same counts (a rank that
    cut its item pass differently would exchange other rows: silent divergence of the replicas) - and follows the budget
    of DESIGN.md section 5: one slice where the exchange is small against the item pass or the backend is gloo, more
    where 2 x 231 MB per step stand against a 2 M-interaction pass."""
    world = 2
    mp.spawn(_auto_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    z = np.load(os.path.join(str(tmp_path), "auto.npz"))
    rule, trainer = z["rule"], z["trainer"]
    assert (rule == rule[0]).all() and (trainer == 1).all()
    r = rule[0]
    assert r[0] == 2 and r[1] == 16                       # configs[2] on 8 GPUs: a 2 M pass (0.2 ms) carries two slices, a 16 M one sixteen
    assert r[2] == 2                                      # configs[1] tables on 8 GPUs at 2 M per rank
    assert r[3] == 1                                      # a 65 536-interaction pass is too short to cut
    assert r[4] == 1                                      # ml-1m's 3706 items: blocks under 1024 rows are not cut
    assert r[6] == 1
    from daisyrec_amd.sharding import auto_exchange_slices
    assert auto_exchange_slices(1_000_000, 64, 1, 1 << 21) == 1 and auto_exchange_slices(1_000_000, 64, 8, 1 << 21, "gloo") == 1
