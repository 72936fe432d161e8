"""The partitioned epoch plan and the staged step (csrc/bpr_staged.hip): index work bit-exact against
the oracle's restatement, the step against oracle.mf_sgd_step (MFRecommender.py:63-97 + SGD) on the
same batches, the phase form against the single call, bitwise reproducibility, id validation."""
import numpy as np
import pytest
import torch

from oracle import bpr_mf_numpy as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _triples(n, U, I, seed, sort=False):
    rng = np.random.default_rng(seed)
    t = np.stack([rng.integers(0, U, n), rng.integers(0, I, n), rng.integers(0, I, n)], 1).astype(np.int32)
    if sort:
        t = t[np.argsort(t[:, 0], kind="stable")]
    return t


def _plan_batches(plan, nb, B):
    out = []
    for k in range(nb):
        u, i, j, ei, es, eu = (t.cpu().numpy() for t in plan.read_batch(k, B))
        out.append((np.stack([u, i, j], 1).astype(np.int64), ei.astype(np.int64), es.astype(np.uint32)))
    return out


@pytest.mark.parametrize("n,B,U,I,sort", [(1000, 64, 300, 200, False), (4097, 256, 300, 200, True),
                                          (513, 1000, 50, 70, False), (7, 1, 5, 5, False),
                                          (70000, 100, 900, 1200, False),        # 700 batches: two LSD passes
                                          (300000, 4096, 20000, 3000, True)])
@pytest.mark.parametrize("tiles", [0, 3])
def test_partitioned_plan_bit_exact(n, B, U, I, sort, tiles, monkeypatch):
    """tiles = 3: at most three tiles (workgroups) per partition, so that a tile walks SEVERAL sub-tiles - what plans
    beyond 67 M records do with the default 16384 tiles - and ends in a partial one"""
    from daisyrec_amd import ops
    if tiles:
        monkeypatch.setenv("DAISY_PART_TILES", str(tiles))
    tri = _triples(n, U, I, n, sort)
    t_dev = torch.from_numpy(tri).to(DEV)
    index = ops.TrainIndex(t_dev, U, I)
    plan = ops.EpochPlan(n, U, I)
    nb = (n + B - 1) // B
    perm = torch.randperm(n, generator=torch.Generator().manual_seed(n))
    inv = np.empty(n, dtype=np.int64)
    inv[perm.numpy()] = np.arange(n)
    cases = [("identity", None, np.arange(n)), ("perm", perm.to(DEV), inv),
             ("feistel", None, O.feistel_positions(n, 99, 3))]
    for order, pm, pos in cases:
        plan.build_indexed(index, B, order=order, perm=pm, seed=99, epoch=3)
        assert plan.num_batches == nb
        samples, spos, ekey, epos = O.partitioned_plan(tri, pos, B)
        got = _plan_batches(plan, nb, B)
        for k in range(nb):
            lo, hi = k * B, min((k + 1) * B, n)
            rows, ei, es = got[k]
            assert np.array_equal(rows, samples[lo:hi]), (order, k)
            # the batch holds exactly the triples the loader would serve at positions [lo, hi)
            assert np.array_equal(np.sort(spos[lo:hi]), np.arange(lo, hi))
            assert np.array_equal(ei, ekey[2 * lo:2 * hi] >> 1), (order, k)
            want_s = ((epos[2 * lo:2 * hi] - lo) | ((ekey[2 * lo:2 * hi] & 1) << 31)).astype(np.uint32)
            assert np.array_equal(es, want_s), (order, k)
    index.close()
    plan.close()


@pytest.mark.parametrize("n,B,U,I", [(1000, 64, 300, 200), (4097, 256, 300, 200), (70000, 100, 900, 1200)])
@pytest.mark.parametrize("tiles", [0, 2])
def test_partitioned_plan_pointwise_bit_exact(n, B, U, I, tiles, monkeypatch):
    """point-wise rows (user, item, label): ONE item entry per row in the static index and in every batch of the plan;
    the label travels in the sample's third column and is not an id (it may exceed nothing: no range check)"""
    from daisyrec_amd import ops
    if tiles:
        monkeypatch.setenv("DAISY_PART_TILES", str(tiles))
    tri = _triples(n, U, I, n + 1)
    tri[:, 2] = np.random.default_rng(n).integers(0, 2, n)              # labels
    tri[::7, 2] = I + 5                                                 # (not an item id: must not be validated as one)
    t_dev = torch.from_numpy(tri).to(DEV)
    index = ops.TrainIndex(t_dev, U, I, pointwise=True)
    plan = ops.EpochPlan(n, U, I)
    nb = (n + B - 1) // B
    pos = O.feistel_positions(n, 4, 2)
    plan.build_indexed(index, B, order="feistel", seed=4, epoch=2)
    samples, spos, ekey, epos = O.partitioned_plan(tri, pos, B, pointwise=True)
    for k in range(nb):
        lo, hi = k * B, min((k + 1) * B, n)
        u, i, j, ei, es, _ = (t.cpu().numpy() for t in plan.read_batch(k, B))
        assert np.array_equal(np.stack([u, i, j], 1), samples[lo:hi]), k
        assert np.array_equal(ei[:hi - lo], ekey[lo:hi] >> 1), k
        assert np.array_equal(es[:hi - lo].astype(np.uint32), (epos[lo:hi] - lo).astype(np.uint32)), k
    index.close()
    plan.close()


def test_index_rejects_out_of_range_ids():
    from daisyrec_amd import ops
    tri = _triples(500, 40, 30, 1)
    for col, bad in ((0, 40), (1, 30), (2, -1), (0, -3)):
        t = tri.copy()
        t[123, col] = bad
        with pytest.raises(ValueError, match="out of range"):
            ops.TrainIndex(torch.from_numpy(t).to(DEV), 40, 30)
    ops.TrainIndex(torch.from_numpy(tri).to(DEV), 40, 30).close()
    # user_base shifts the accepted window (user-sharded tables)
    t = tri.copy()
    t[:, 0] += 1000
    ops.TrainIndex(torch.from_numpy(t).to(DEV), 40, 30, user_base=1000).close()
    with pytest.raises(ValueError, match="out of range"):
        ops.TrainIndex(torch.from_numpy(t).to(DEV), 40, 30, user_base=0)


def _tables(U, I, d, seed, scale=0.1):
    rng = np.random.default_rng(seed)
    return ((rng.standard_normal((U, d)) * scale).astype(np.float32),
            (rng.standard_normal((I, d)) * scale).astype(np.float32))


@pytest.mark.parametrize("d", [64, 32, 8, 20, 128, 100, 7, 256])
@pytest.mark.parametrize("loss", ["BPR", "HL", "TL", "CL", "SL"])
@pytest.mark.parametrize("sparse,merge", [(None, None), ("1", None), (None, "0"), ("1", "0")])
def test_staged_epoch_matches_oracle(d, loss, sparse, merge, monkeypatch):
    """A whole epoch through the partitioned plan + fit_epoch_sgd(fused), replayed by the oracle on the
    batches the plan serves: hot users/items, runs that cross lane groups and chunks, a partial batch.
    CL / SL: point-wise rows (user, item, label), one entry per row.
    sparse = "1": the sparse flavour of the item pass (runs of 4 entries, the Q row of every entry gathered with its
    staged row) forced onto these DENSE batches - long segments across runs and chunks, which it meets rarely where
    the library picks it by itself."""
    from daisyrec_amd import ops
    if sparse:
        monkeypatch.setenv("DAISY_STAGED_SPARSE", sparse)
    if merge:        # "0": four launches per step; default at this size: three (the item launch carries the user pass's edge work)
        monkeypatch.setenv("DAISY_STAGED_MERGE", merge)
    U, I, n, B = 37, 23, 1500, 400         # heavy collisions: every row is shared inside a batch
    tri = _triples(n, U, I, d)
    point = loss in ("CL", "SL")
    if point:
        tri[:, 2] = np.random.default_rng(d).integers(0, 2, n)
    P0, Q0 = _tables(U, I, d, d + 1)
    t_dev = torch.from_numpy(tri).to(DEV)
    index, plan = ops.TrainIndex(t_dev, U, I, pointwise=point), ops.EpochPlan(n, U, I)
    plan.build_indexed(index, B, order="feistel", seed=5, epoch=1)
    nb = plan.num_batches
    P, Q = torch.from_numpy(P0).to(DEV), torch.from_numpy(Q0).to(DEV)
    ctx = ops.BprContext(B, d, U, I)
    sl = torch.zeros(nb, dtype=torch.float64, device=DEV)
    lid = ops.LOSS_IDS[loss]
    ctx.fit_epoch_sgd(plan, P, Q, 0.05, 1e-3, 2e-3, loss_type=lid, item_mode=ops.ITEM_MODES["fused"], step_losses=sl)
    torch.cuda.synchronize()
    Pn, Qn = P0.astype(np.float64), Q0.astype(np.float64)
    for k, (rows, _, _) in enumerate(_plan_batches(plan, nb, B)):
        want, Pn, Qn = O.mf_sgd_step(Pn, Qn, rows[:, 0], rows[:, 1], rows[:, 2], 0.05, 1e-3, 2e-3, loss_type=lid)
        assert abs(float(sl[k].cpu()) - want) <= 1e-5 * abs(want), (k, float(sl[k].cpu()), want)
    assert np.abs(P.cpu().numpy() - Pn).max() < 5e-6 and np.abs(Q.cpu().numpy() - Qn).max() < 5e-6
    ctx.close(); plan.close(); index.close()


@pytest.mark.parametrize("d,B,I", [(64, 3000, 9000), (32, 1024, 4000), (100, 777, 50000), (64, 6000, 100000)])
@pytest.mark.parametrize("sparse", [None, "0"])
def test_sparse_batches_match_the_oracle(d, B, I, sparse, monkeypatch):
    """Few entries per item - what every batch of the reference's size range (basic.yaml:23: 256 ... a few thousand) is
    over a real item table: the library takes the sparse flavour of the item pass by itself (sparse=None); "0" forces
    the dense one onto the same batches (its LDS window covers none of their rows).  Two steps against the oracle."""
    from daisyrec_amd import ops
    if sparse:
        monkeypatch.setenv("DAISY_STAGED_SPARSE", sparse)
    U, n = 5000, 2 * B
    rng = np.random.default_rng(d + B)
    tri = np.stack([rng.integers(0, U, n), rng.integers(0, I, n), rng.integers(0, I, n)], 1).astype(np.int32)
    tri[:40, 1] = 7                        # one segment across runs even here
    P0, Q0 = _tables(U, I, d, d + 3)
    t_dev = torch.from_numpy(tri).to(DEV)
    index, plan = ops.TrainIndex(t_dev, U, I), ops.EpochPlan(n, U, I)
    plan.build_indexed(index, B, order="feistel", seed=3, epoch=1)
    P, Q = torch.from_numpy(P0).to(DEV), torch.from_numpy(Q0).to(DEV)
    ctx = ops.BprContext(B, d, U, I)
    sl = torch.zeros(2, dtype=torch.float64, device=DEV)
    ctx.fit_epoch_sgd(plan, P, Q, 0.05, 1e-3, 2e-3, item_mode=ops.ITEM_MODES["fused"], step_losses=sl)
    torch.cuda.synchronize()
    Pn, Qn = P0.astype(np.float64), Q0.astype(np.float64)
    for k, (rows, _, _) in enumerate(_plan_batches(plan, 2, B)):
        want, Pn, Qn = O.mf_sgd_step(Pn, Qn, rows[:, 0], rows[:, 1], rows[:, 2], 0.05, 1e-3, 2e-3)
        assert abs(float(sl[k].cpu()) - want) <= 1e-5 * abs(want), (k, float(sl[k].cpu()), want)
    assert np.abs(P.cpu().numpy() - Pn).max() < 5e-6 and np.abs(Q.cpu().numpy() - Qn).max() < 5e-6
    ctx.close(); plan.close(); index.close()


@pytest.mark.parametrize("d", [32, 64, 100])
@pytest.mark.parametrize("opt", ["sgd", "adam"])
def test_p_stream_user_rows_match_the_oracle(d, opt):
    """StreamView::p_stream - user rows read with nontemporal loads and written with nontemporal owner stores (what the
    host switches on for user tables beyond 512 MB, i.e. the code path of bench.py's `secondary` number) - forced on
    through daisy_bpr_ctx_set_p_stream on heavily colliding batches: only the cache policy may differ, so the epoch must
    equal the oracle's (MFRecommender.py:63-97) like the default path does, and the default path bit for bit."""
    from daisyrec_amd import ops
    from daisyrec_amd.model.AbstractRecommender import _AdamState
    U, I, n, B = 37, 23, 1500, 400
    tri = _triples(n, U, I, d + 11)
    tri[:300, 0] = 5                       # a run that crosses chunks (edge records: their owners store too)
    P0, Q0 = _tables(U, I, d, d + 12)
    t_dev = torch.from_numpy(tri).to(DEV)
    index, plan = ops.TrainIndex(t_dev, U, I), ops.EpochPlan(n, U, I)
    plan.build_indexed(index, B, order="feistel", seed=5, epoch=1)
    nb = plan.num_batches
    batches = _plan_batches(plan, nb, B)
    lid, lr = ops.LOSS_IDS["BPR"], (0.05 if opt == "sgd" else 0.01)
    outs = {}
    for mode in (False, True):
        P, Q = torch.from_numpy(P0).to(DEV), torch.from_numpy(Q0).to(DEV)
        ctx = ops.BprContext(B, d, U, I)
        ctx.set_p_stream(mode)
        sl = torch.zeros(nb, dtype=torch.float64, device=DEV)
        if opt == "sgd":
            ctx.fit_epoch_sgd(plan, P, Q, lr, 1e-3, 2e-3, loss_type=lid, item_mode=ops.ITEM_MODES["fused"], step_losses=sl)
        else:
            adam = _AdamState(P, Q, lr, None, kind="adam", max_steps=4)
            for k in range(nb):
                ctx.set_batch_from_plan(plan, k)
                adam.step(ctx, P, Q, 1e-3, 2e-3, lid, ops.ITEM_MODES["fused"])
                sl[k] = ctx.stats[7]
            adam.flush()
        torch.cuda.synchronize()
        outs[mode] = (P.cpu().numpy(), Q.cpu().numpy(), sl.cpu().numpy())
        ctx.close()
    assert np.array_equal(outs[True][0], outs[False][0]) and np.array_equal(outs[True][1], outs[False][1])
    assert np.array_equal(outs[True][2], outs[False][2])
    Pn, Qn = P0.astype(np.float64), Q0.astype(np.float64)
    ref = O.DenseAdam([P0.shape, Q0.shape], lr)
    for k, (rows, _, _) in enumerate(batches):
        if opt == "sgd":
            want, Pn, Qn = O.mf_sgd_step(Pn, Qn, rows[:, 0], rows[:, 1], rows[:, 2], lr, 1e-3, 2e-3, loss_type=lid)
        else:
            want, gP, gQ = O.mf_pair_grad(Pn, Qn, rows[:, 0], rows[:, 1], rows[:, 2], 1e-3, 2e-3, lid)
            Pn, Qn = ref.step([Pn, Qn], [gP, gQ])
        assert abs(outs[True][2][k] - want) <= 2e-5 * abs(want), (k, outs[True][2][k], want)
    for got, want_t in ((outs[True][0], Pn), (outs[True][1], Qn)):
        diff = np.abs(got - want_t)
        if opt == "sgd":
            assert diff.max() < 5e-6
        else:       # (Adam: see test_staged_adam_epochs_match_the_dense_oracle for the tolerance)
            assert (diff > 2e-5).mean() < 2e-3 and np.median(diff) < 1e-7
    plan.close(); index.close()


def test_p_stream_step_at_ten_million_users():
    """The regime the switch exists for: a 10 M x 64 user table (2.56 GB, ten Infinity Caches) - one 1 M-sample staged
    step with the AUTOMATIC setting (on by table size) against a plain torch fp32 restatement of the closed form on the
    GPU, like test_c2_scale_step_properties; then the same step with the switch forced off: identical bits."""
    from daisyrec_amd import ops
    U, I, d, B = 10_000_000, 1_000_000, 64, 1 << 20
    gen = torch.Generator(device=DEV)
    gen.manual_seed(7)
    P = torch.randn(U, d, device=DEV, generator=gen) * 0.01
    Q = torch.randn(I, d, device=DEV, generator=gen) * 0.01
    u = torch.randint(0, U, (B,), device=DEV, generator=gen, dtype=torch.int32)
    u[:50_000] = u[0]                      # a run across many chunks
    i = torch.randint(0, I, (B,), device=DEV, generator=gen, dtype=torch.int32)
    j = torch.randint(0, I, (B,), device=DEV, generator=gen, dtype=torch.int32)
    lr, r1, r2 = 0.01, 1e-3, 1e-3
    ul, il, jl = u.long(), i.long(), j.long()
    pu, qi, qj = P[ul], Q[il], Q[jl]
    x = (pu * qi).sum(-1) - (pu * qj).sum(-1)
    s = torch.sigmoid(x)
    loss = -(1e-10 + s).log().double().sum()
    nU, nI, nJ = (t.double().pow(2).sum().sqrt() for t in (pu, qi, qj))
    loss = loss + r1 * (pu.abs().double().sum() + qi.abs().double().sum() + qj.abs().double().sum()) + r2 * (nU + nI + nJ)
    c = (-(s * (1 - s)) / (1e-10 + s)).unsqueeze(1)
    dP = -lr * (c * (qi - qj) + r1 * pu.sign() + r2 * pu / nU.float())
    Qn = Q.clone()
    Qn.index_add_(0, il, -lr * (c * pu + r1 * qi.sign() + r2 * qi / nI.float()))
    Qn.index_add_(0, jl, -lr * (-c * pu + r1 * qj.sign() + r2 * qj / nJ.float()))
    del qi, qj, x, s
    res = []
    for mode in ("auto", False):
        P1, Q1 = P.clone(), Q.clone()
        ctx = ops.BprContext(B, d, U, I)
        ctx.set_p_stream(mode)
        sl = torch.zeros(1, dtype=torch.float64, device=DEV)
        ctx.set_batch(u, i, j)
        ctx.sgd_step(P1, Q1, lr, r1, r2, item_mode=ops.ITEM_MODES["fused"], step_loss=sl)
        torch.cuda.synchronize()
        assert abs(float(sl.cpu()) - float(loss.cpu())) <= 1e-5 * float(loss.cpu())
        res.append((P1, Q1))
        ctx.close()
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    P1, Q1 = res[0]
    assert float((Q1 - Qn).abs().max().cpu()) < 1e-6
    Pn = P.clone()
    Pn.index_add_(0, ul, dP)
    dPmax = (P1 - Pn).abs()
    long_user = int(ul[0])                 # (its 50 000 terms of ~1e-4 meet in another order: a few fp32 ulps of the sum)
    assert float(dPmax[long_user].max().cpu()) < 2e-5
    dPmax[long_user] = 0
    assert float(dPmax.max().cpu()) < 1e-6
    untouched = torch.ones(U, dtype=torch.bool, device=DEV)
    untouched[ul] = False
    assert torch.equal(P1[untouched], P[untouched])


@pytest.mark.parametrize("d,loss", [(64, "BPR"), (32, "TL"), (100, "BPR"), (64, "CL"), (20, "SL"), (128, "HL")])
def test_staged_adam_epochs_match_the_dense_oracle(d, loss):
    """torch.optim.Adam applied by the row owners of the staged step (lazy: catch-up of the referenced rows, step t on the
    rows with a gradient, flush at the epoch's end) against oracle.DenseAdam - every row of both tables in every step -
    over two epochs of colliding batches: runs across lane groups and chunks (the slot finishers and the edge kernels
    apply Adam too), rows that skip steps (the second epoch's plan differs), a partial batch."""
    from daisyrec_amd import ops
    from daisyrec_amd.model.AbstractRecommender import _AdamState
    U, I, n, B = 61, 43, 1300, 400
    tri = _triples(n, U, I, d + 3)
    tri[:300, 0] = 5                       # a run that crosses chunks
    tri[300:330, 1] = 7
    point = loss in ("CL", "SL")
    if point:
        tri[:, 2] = np.random.default_rng(d).integers(0, 2, n)
    P0, Q0 = _tables(U, I, d, d + 2)
    t_dev = torch.from_numpy(tri).to(DEV)
    index, plan = ops.TrainIndex(t_dev, U, I, pointwise=point), ops.EpochPlan(n, U, I)
    P, Q = torch.from_numpy(P0).to(DEV), torch.from_numpy(Q0).to(DEV)
    ctx = ops.BprContext(B, d, U, I)
    lid, lr = ops.LOSS_IDS[loss], 0.01
    adam = _AdamState(P, Q, lr, None, kind="adam", max_steps=4)        # (the table of step constants grows on demand)
    ref = O.DenseAdam([P0.shape, Q0.shape], lr)
    Pn, Qn = P0.copy(), Q0.copy()
    for epoch in (1, 2):
        plan.build_indexed(index, B, order="feistel", seed=9, epoch=epoch)
        for k, (rows, _, _) in enumerate(_plan_batches(plan, plan.num_batches, B)):
            ctx.set_batch_from_plan(plan, k)
            adam.step(ctx, P, Q, 1e-3, 2e-3, lid, ops.ITEM_MODES["fused"])
            if point:
                want, gP, gQ = O.mf_point_grad(Pn, Qn, rows[:, 0], rows[:, 1], rows[:, 2], 1e-3, 2e-3, lid)
            else:
                want, gP, gQ = O.mf_pair_grad(Pn, Qn, rows[:, 0], rows[:, 1], rows[:, 2], 1e-3, 2e-3, lid)
            Pn, Qn = ref.step([Pn, Qn], [gP, gQ])
            got = float(ctx.stats[7].cpu())
            assert abs(got - want) <= 2e-5 * abs(want), (epoch, k, got, want)
        adam.flush()
        torch.cuda.synchronize()
        # Adam divides by sqrt(v): where a gradient element is itself round-off sized (the data term cancelling the
        # regulariser) its fp32 noise decides a step of up to lr in BOTH implementations - a handful of the ~10^4
        # elements; everything else agrees to fp32 round-off
        for got, want_t in ((P.cpu().numpy(), Pn), (Q.cpu().numpy(), Qn)):
            diff = np.abs(got - want_t)
            assert (diff > 2e-5).mean() < 2e-3 and diff.max() < 0.5 * lr * 2 * epoch * plan.num_batches, (epoch, diff.max())
            assert np.median(diff) < 1e-7
    ctx.close(); plan.close(); index.close()


@pytest.mark.parametrize("d,loss", [(64, "BPR"), (32, "TL"), (100, "CL"), (64, "SL")])
def test_staged_fm_epoch_matches_oracle(d, loss):
    """FM's biases on the staged step (plain / point flavours): a whole epoch through the partitioned plan against
    oracle.fm_numpy.fm_sgd_step - the user bias along its run, the item bias along its segment, bias_ from the batch sum"""
    from daisyrec_amd import ops
    from oracle import fm_numpy as F
    U, I, n, B = 37, 23, 1500, 400
    tri = _triples(n, U, I, d + 5)
    tri[:200, 0] = 3
    point = loss in ("CL", "SL")
    if point:
        tri[:, 2] = np.random.default_rng(d).integers(0, 2, n)
    P0, Q0 = _tables(U, I, d, d + 4)
    rng = np.random.default_rng(d)
    bu0, bi0 = (rng.standard_normal(U) * 0.1).astype(np.float32), (rng.standard_normal(I) * 0.1).astype(np.float32)
    b00 = np.array([0.05], np.float32)
    t_dev = torch.from_numpy(tri).to(DEV)
    index, plan = ops.TrainIndex(t_dev, U, I, pointwise=point), ops.EpochPlan(n, U, I)
    plan.build_indexed(index, B, order="feistel", seed=5, epoch=1)
    nb = plan.num_batches
    w = [torch.from_numpy(x.copy()).to(DEV) for x in (P0, Q0, bu0, bi0, b00)]
    ctx = ops.BprContext(B, d, U, I)
    ctx.set_bias(w[2], w[3], w[4], g_i_bias=torch.zeros(I, device=DEV))
    sl = torch.zeros(nb, dtype=torch.float64, device=DEV)
    lid = ops.LOSS_IDS[loss]
    lr = 0.002 if loss == "SL" else 0.05         # (the squared loss diverges at 0.05 on these collision-heavy batches)
    ctx.fit_epoch_sgd(plan, w[0], w[1], lr, 1e-3, 2e-3, loss_type=lid, item_mode=ops.ITEM_MODES["fused"], step_losses=sl)
    torch.cuda.synchronize()
    cur = [P0, Q0, bu0, bi0, b00]
    for k, (rows, _, _) in enumerate(_plan_batches(plan, nb, B)):
        want, *cur = F.fm_sgd_step(*cur, rows[:, 0], rows[:, 1], rows[:, 2], lr, 1e-3, 2e-3, loss_type=lid)
        assert abs(float(sl[k].cpu()) - want) <= 1e-5 * abs(want), (k, float(sl[k].cpu()), want)
    for got, want in zip(w, cur):
        assert np.abs(got.cpu().numpy().reshape(-1) - np.asarray(want).reshape(-1)).max() < 6e-6
    ctx.close(); plan.close(); index.close()


@pytest.mark.parametrize("merge", [None, "1"])
def test_staged_large_batch_against_chunked_and_reproducible(merge, monkeypatch):
    """Throughput shapes (runs across chunks on both sides): the staged step equals the phase kernels to
    round-off, touches the same rows, and two runs give identical bits.  merge = "1": the three-launch form forced onto
    this 131 072-sample batch (2048 rows of partial sums for every item workgroup to add; by itself the library stops
    at 512)."""
    from daisyrec_amd import ops
    if merge:
        monkeypatch.setenv("DAISY_STAGED_MERGE", merge)
    U, I, d, n, B = 20000, 3000, 64, 400000, 131072
    tri = _triples(n, U, I, 3, sort=True)
    tri[:5000, 0] = 7                      # one very long user run
    tri[5000:9000, 1] = 11                 # one very hot item
    tri = tri[np.argsort(tri[:, 0], kind="stable")]
    t_dev = torch.from_numpy(tri).to(DEV)
    P0, Q0 = _tables(U, I, d, 9, 0.05)
    index, plan, plan0 = ops.TrainIndex(t_dev, U, I), ops.EpochPlan(n, U, I), ops.EpochPlan(n, U, I)
    plan.build_indexed(index, B, order="feistel", seed=1, epoch=0)
    plan0.build(t_dev, B, order="feistel", seed=1, epoch=0, user_sorted=True)
    outs = []
    for mode, pl in (("fused", plan), ("fused", plan), ("chunked", plan0), ("fused", plan0)):
        P, Q = torch.from_numpy(P0).to(DEV), torch.from_numpy(Q0).to(DEV)
        ctx = ops.BprContext(B, d, U, I)
        ctx.fit_epoch_sgd(pl, P, Q, 0.01, 1e-3, 1e-3, item_mode=ops.ITEM_MODES[mode])
        torch.cuda.synchronize()
        outs.append((P.cpu().numpy(), Q.cpu().numpy(), float(ctx.epoch_acc[0].cpu())))
        ctx.close()
    a, b, c, e = outs
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2]      # reproducible
    for x in (c, e):                                                                    # same step, other kernels / layout
        assert abs(a[2] - x[2]) <= 1e-6 * abs(x[2])
        assert np.abs(a[0] - x[0]).max() < 2e-6 and np.abs(a[1] - x[1]).max() < 2e-6
    # the same rows move (a row is untouched iff every element kept its bits)
    assert np.array_equal((a[0] != P0).any(1), (c[0] != P0).any(1))
    assert np.array_equal((a[1] != Q0).any(1), (c[1] != Q0).any(1))
    plan.close(); plan0.close(); index.close()


def test_staged_phases_equal_single_call():
    """prenorm -> user -> finalize -> item(gQ, cnt) -> apply_counts (the multi-GPU form) = sgd_step(fused)."""
    from daisyrec_amd import ops
    U, I, d, B = 500, 300, 64, 5000
    tri = _triples(B, U, I, 4)
    P0, Q0 = _tables(U, I, d, 5)
    u, i, j = (torch.from_numpy(tri[:, c].copy()).to(DEV) for c in range(3))
    res = []
    for phased in (False, True):
        P, Q = torch.from_numpy(P0).to(DEV), torch.from_numpy(Q0).to(DEV)
        ctx = ops.BprContext(B, d, U, I)
        ctx.set_batch(u, i, j)
        if not phased:
            ctx.sgd_step(P, Q, 0.02, 1e-3, 1e-3, item_mode=ops.ITEM_MODES["fused"])
        else:
            cnt = torch.zeros(I, 2, device=DEV)
            ctx.staged_prenorm(P)
            ctx.staged_user(P, Q, 0.02, 1e-3, 1e-3)
            ctx.finalize(1e-3, 1e-3)
            ctx.staged_item(0.02, 1e-3, 1e-3, gQ=ctx.gQ, cnt=cnt)
            ops.item_apply_counts(Q, ctx.gQ, cnt, 0.02, 1e-3, 1e-3, ctx.stats)
            torch.cuda.synchronize()
            assert float(ctx.gQ.abs().max().cpu()) == 0.0 and float(cnt.abs().max().cpu()) == 0.0
        torch.cuda.synchronize()
        res.append((P.cpu().numpy(), Q.cpu().numpy(), float(ctx.stats[7].cpu())))
        ctx.close()
    assert np.array_equal(res[0][0], res[1][0]) and res[0][2] == res[1][2]
    assert np.abs(res[0][1] - res[1][1]).max() < 1e-7
    want, Pn, Qn = O.mf_sgd_step(P0, Q0, tri[:, 0], tri[:, 1], tri[:, 2], 0.02, 1e-3, 1e-3)
    assert abs(res[0][2] - want) <= 1e-5 * abs(want)
    assert np.abs(res[0][0] - Pn).max() < 3e-6 and np.abs(res[0][1] - Qn).max() < 3e-6


@pytest.mark.parametrize("blocks", [None, "1"])
@pytest.mark.parametrize("loss", ["BPR", "TL"])
def test_item_pass_in_slices_equals_the_whole_pass(loss, blocks, monkeypatch):
    """daisy_bpr_staged_item_slices / _slice: the item pass cut at item boundaries (what a multi-GPU step pipelines
    its exchange under) writes the same counts and the same gQ up to the order in which a long segment's partial sums
    meet (the cuts move the workgroup boundaries) - even cuts, cuts at popular items, empty slices.  blocks = "1": the
    edge chains of every slice in two levels (round 5) - the gradient form, slice-relative block numbering."""
    from daisyrec_amd import ops
    if blocks:
        monkeypatch.setenv("DAISY_EDGE_BLOCKS", blocks)
    U, I, d, B = 400, 257, 64, 20000
    rng = np.random.default_rng(3)
    tri = np.stack([rng.integers(0, U, B), (rng.zipf(1.3, B) % I), rng.integers(0, I, B)], 1).astype(np.int32)
    tri[:, 1][tri[:, 1] >= 200] = 7                         # nothing positive in [200, 257); item 7 very popular
    P0, Q0 = _tables(U, I, d, 9)
    u, i, j = (torch.from_numpy(tri[:, c].copy()).to(DEV) for c in range(3))
    lid = ops.LOSS_IDS[loss]
    P, Q = torch.from_numpy(P0).to(DEV), torch.from_numpy(Q0).to(DEV)
    ctx = ops.BprContext(B, d, U, I)
    ctx.set_batch(u, i, j)
    ctx.staged_prenorm(P)
    ctx.staged_user(P, Q, 0.02, 1e-3, 1e-3, lid)
    ctx.finalize(1e-3, 1e-3)
    g0, c0 = torch.zeros(I, d, device=DEV), torch.zeros(I, 2, device=DEV)
    ctx.staged_item(0.02, 1e-3, 1e-3, gQ=g0, cnt=c0, loss_type=lid)
    for bounds in ([0, I], [0, 64, 128, 192, I], [0, 7, 8, 8, 100, 300], list(range(0, 16 * 17, 17))[:16] + [I],
                   [0, 1, 2, 3, I + 5]):
        g1, c1 = torch.zeros(I, d, device=DEV), torch.zeros(I, 2, device=DEV)
        ctx.staged_item_slices(bounds)
        for s_ in reversed(range(len(bounds) - 1)):         # any order
            ctx.staged_item_slice(s_, 0.02, 1e-3, 1e-3, g1, c1, loss_type=lid)
        torch.cuda.synchronize()
        assert torch.equal(c0, c1), bounds
        scale = float(g0.abs().max().cpu())
        assert float((g0 - g1).abs().max().cpu()) <= 2e-6 * scale, bounds
        if len(bounds) == 2:
            assert torch.equal(g0, g1)                      # one slice = the whole pass
        # a slice touches only its own rows
        g2, c2 = torch.zeros(I, d, device=DEV), torch.zeros(I, 2, device=DEV)
        ctx.staged_item_slice(0, 0.02, 1e-3, 1e-3, g2, c2, loss_type=lid)
        torch.cuda.synchronize()
        hi = min(bounds[1], I)
        assert float(g2[hi:].abs().max().cpu() if hi < I else 0.0) == 0.0
        assert float((g2[:hi] - g0[:hi]).abs().max().cpu()) <= 2e-6 * scale
    with pytest.raises(ValueError, match="cover"):
        ctx.staged_item_slices([0, 10, 20])
    with pytest.raises(ValueError, match="slice"):
        ctx.set_batch(u, i, j)
        ctx.staged_item_slice(0, 0.02, 1e-3, 1e-3, g0, c0)
    ctx.close()


def test_norm_cache_follows_torch_side_edits():
    """P changed by a torch op between two staged steps: the wrapper sees the version counter move and the
    row-norm cache is rebuilt (a stale cache would put the wrong |P[u]|_F into the user update)."""
    from daisyrec_amd import ops
    U, I, d, B = 200, 100, 32, 1000
    tri = _triples(B, U, I, 8)
    P0, Q0 = _tables(U, I, d, 2)
    u, i, j = (torch.from_numpy(tri[:, c].copy()).to(DEV) for c in range(3))
    P, Q = torch.from_numpy(P0).to(DEV), torch.from_numpy(Q0).to(DEV)
    ctx = ops.BprContext(B, d, U, I)
    ctx.set_batch(u, i, j)
    ctx.sgd_step(P, Q, 0.02, 0.0, 0.5, item_mode=ops.ITEM_MODES["fused"])
    P.mul_(3.0)                                           # torch-side edit
    P1, Q1 = P.cpu().numpy().copy(), Q.cpu().numpy().copy()
    ctx.set_batch(u, i, j)
    ctx.sgd_step(P, Q, 0.02, 0.0, 0.5, item_mode=ops.ITEM_MODES["fused"])
    torch.cuda.synchronize()
    _, Pn, Qn = O.mf_sgd_step(P1, Q1, tri[:, 0], tri[:, 1], tri[:, 2], 0.02, 0.0, 0.5)
    assert np.abs(P.cpu().numpy() - Pn).max() < 5e-6 and np.abs(Q.cpu().numpy() - Qn).max() < 5e-6
    ctx.close()


def test_partitioned_plan_refused_by_phase_kernels():
    from daisyrec_amd import ops
    U, I, d, n = 50, 40, 16, 300
    t_dev = torch.from_numpy(_triples(n, U, I, 1)).to(DEV)
    index, plan = ops.TrainIndex(t_dev, U, I), ops.EpochPlan(n, U, I)
    plan.build_indexed(index, 100)
    ctx = ops.BprContext(100, d, U, I)
    P, Q = (torch.zeros(x, d, device=DEV) for x in (U, I))
    ctx.set_batch_from_plan(plan, 0)
    with pytest.raises(RuntimeError, match="partitioned plan"):
        ctx.forward(P, Q)
    with pytest.raises(RuntimeError):
        ctx.sgd_step(P, Q, 0.1, 0, 0, item_mode=ops.ITEM_MODES["chunked"])
    ctx.close(); plan.close(); index.close()


@pytest.mark.parametrize("d,B,loss", [(32, 256, "BPR"), (64, 256, "HL"), (20, 100, "TL"), (128, 100, "BPR"), (8, 1, "BPR"), (256, 50, "BPR"), (100, 77, "BPR"),
                                      (100, 256, "BPR"), (128, 256, "HL"), (72, 256, "TL")])      # only the user rows fit the LDS
def test_small_batch_epoch_in_one_workgroup(d, B, loss):
    """B <= 256 over the sorted plan: fit_epoch_sgd runs every step of the epoch inside one persistent
    workgroup (csrc/bpr_small.hip).  Same users and items recur from step to step, so a stale cache line
    between the phases or the steps would show; checked against the oracle stepping through the same batches."""
    from daisyrec_amd import ops
    U, I, n = 60, 45, 2600 if B > 1 else 40
    tri = _triples(n, U, I, d + B)
    P0, Q0 = _tables(U, I, d, B)
    t_dev = torch.from_numpy(tri).to(DEV)
    plan = ops.EpochPlan(n, U, I).build(t_dev, B, order="feistel", seed=3, epoch=2)
    nb = plan.num_batches
    lid = ops.LOSS_IDS[loss]
    res = []
    for env_small in (True, False):
        P, Q = torch.from_numpy(P0).to(DEV), torch.from_numpy(Q0).to(DEV)
        ctx = ops.BprContext(B, d, U, I)
        sl = torch.zeros(nb, dtype=torch.float64, device=DEV)
        mode = ops.ITEM_MODES["fused" if env_small else "sorted"]       # 'sorted' keeps the per-step phase kernels
        ctx.fit_epoch_sgd(plan, P, Q, 0.05, 1e-3, 2e-3, loss_type=lid, item_mode=mode, step_losses=sl)
        torch.cuda.synchronize()
        res.append((P.cpu().numpy(), Q.cpu().numpy(), sl.cpu().numpy(), float(ctx.epoch_acc[0].cpu()),
                    float(ctx.stats[7].cpu())))
        ctx.close()
    Pn, Qn = P0.astype(np.float64), Q0.astype(np.float64)
    want = []
    for k in range(nb):
        u, i, j = (t.cpu().numpy().astype(np.int64) for t in plan.read_batch(k, B)[:3])
        w, Pn, Qn = O.mf_sgd_step(Pn, Qn, u, i, j, 0.05, 1e-3, 2e-3, loss_type=lid)
        want.append(w)
    for P, Q, sl, acc, last in res:
        np.testing.assert_allclose(sl, want, rtol=1e-5)
        assert abs(acc - sum(want)) <= 1e-5 * abs(sum(want)) and abs(last - want[-1]) <= 1e-5 * abs(want[-1])
        tol = 1e-5 * max(1.0, d / 64)            # fp32 dot products over d terms, nine collision-heavy steps
        assert np.abs(P - Pn).max() < tol and np.abs(Q - Qn).max() < tol
    assert np.abs(res[0][0] - res[1][0]).max() < 2e-6 and np.abs(res[0][1] - res[1][1]).max() < 2e-6
    plan.close()


@pytest.mark.parametrize("d,loss,B", [(64, "BPR", 400), (32, "CL", 256), (100, "HL", 1300)])
def test_adam_epoch_in_one_call_equals_the_step_by_step_loop(d, loss, B):
    """daisy_bpr_fit_epoch_adam (ABI 6; the loop of AbstractRecommender.py:118-128 with torch.optim.Adam, one enqueue per
    epoch) against the same epochs driven step by step through daisy_bpr_staged_adam_step + the flush: both tables, both
    moments, the stamps, the epoch accumulator and every step loss bit for bit, over two epochs (the second starts from
    step nb + 1 with another plan)."""
    from daisyrec_amd import ops
    U, I, n = 61, 43, 1300
    tri = _triples(n, U, I, d + 5)
    tri[:300, 0] = 5
    point = loss in ("CL", "SL")
    if point:
        tri[:, 2] = np.random.default_rng(d).integers(0, 2, n)
    P0, Q0 = _tables(U, I, d, d + 1)
    t_dev = torch.from_numpy(tri).to(DEV)
    lid = ops.LOSS_IDS[loss]
    res = []
    for one_call in (False, True):
        index, plan = ops.TrainIndex(t_dev, U, I, pointwise=point), ops.EpochPlan(n, U, I)
        P, Q = torch.from_numpy(P0).to(DEV), torch.from_numpy(Q0).to(DEV)
        ctx = ops.BprContext(B, d, U, I)
        adam = ops.LazyAdam(P, Q, 0.01, 2)                 # (a table of 2 steps: both paths have to grow it)
        losses, accs = [], []
        for epoch in (1, 2):
            plan.build_indexed(index, B, order="feistel", seed=4, epoch=epoch)
            nb = plan.num_batches
            sl = torch.zeros(nb, dtype=torch.float64, device=DEV)
            ctx.epoch_acc.zero_()
            if one_call:
                adam.fit_epoch(ctx, plan, 1e-3, 2e-3, lid, step_losses=sl)
            else:
                for k in range(nb):
                    ctx.set_batch_from_plan(plan, k)
                    adam.staged_step(ctx, 1e-3, 2e-3, lid, step_loss=sl[k:k + 1])
                adam.flush()
            torch.cuda.synchronize()
            losses.append(sl.clone())
            accs.append(ctx.epoch_acc.clone())
        assert adam.t == 2 * nb
        res.append((P.clone(), Q.clone(), [x.clone() for x in adam.m + adam.v + adam.last], losses, accs))
        ctx.close(); plan.close(); index.close()
    a, b = res
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert all(torch.equal(x, y) for x, y in zip(a[2], b[2]))
    assert all(torch.equal(x, y) for x, y in zip(a[3], b[3])) and all(torch.equal(x, y) for x, y in zip(a[4], b[4]))
    assert float(a[4][1][0].cpu()) > 0 and float(a[4][1][1].cpu()) == 0


@pytest.mark.parametrize("d,loss,B,U,I", [(32, "BPR", 256, 943, 1152), (64, "HL", 200, 61, 43), (100, "TL", 256, 300, 200),
                                          (8, "BPR", 1, 5, 7)])
def test_small_batch_adam_epoch_in_one_persistent_workgroup(d, loss, B, U, I):
    """Round 6: torch.optim.Adam at the reference's batch size (basic.yaml:23, B = 256) - every step of an epoch inside the
    persistent workgroup of csrc/bpr_small.hip (catch-up of the step's distinct rows, forward, owner-applied Adam), reached
    through daisy_bpr_fit_epoch_adam on a SORTED plan.  Against the oracle's dense Adam on the same batches (every row of
    both tables steps in every step: AbstractRecommender.py:54,119-126): epoch losses, both tables after the flush, over two
    epochs; the three LDS staging modes (d = 32 / 64: all rows staged, d = 100: user rows only); hot users (one user in 300
    samples); a batch of one."""
    from daisyrec_amd import ops
    n = 1500 if B > 1 else 7
    tri = _triples(n, U, I, d + 11)
    tri[:min(300, n // 2), 0] = 3
    tri = tri[np.argsort(tri[:, 0], kind="stable")]
    P0, Q0 = _tables(U, I, d, d + 2)
    lid = ops.LOSS_IDS[loss]
    lr, r1, r2 = 0.01, 1e-3, 2e-3
    t_dev = torch.from_numpy(tri).to(DEV)
    P, Q = torch.from_numpy(P0).to(DEV), torch.from_numpy(Q0).to(DEV)
    ctx, plan = ops.BprContext(B, d, U, I), ops.EpochPlan(n, U, I)
    adam = ops.LazyAdam(P, Q, lr, 2)
    ref = O.DenseAdam([P0.shape, Q0.shape], lr)
    Pr, Qr = P0.astype(np.float64), Q0.astype(np.float64)
    for epoch in (1, 2):
        plan.build(t_dev, B, order="feistel", seed=9, epoch=epoch, user_sorted=True)
        assert ops.LazyAdam.small_epoch_supported(ctx, plan, lid)
        nb = plan.num_batches
        sl = torch.zeros(nb, dtype=torch.float64, device=DEV)
        ctx.epoch_acc.zero_()
        adam.fit_epoch(ctx, plan, r1, r2, lid, step_losses=sl)
        torch.cuda.synchronize()
        want = []
        for k in range(nb):
            u, i, j = (x.cpu().numpy().astype(np.int64) for x in plan.read_batch(k, B)[:3])
            loss_k, gP, gQ = O.mf_pair_grad(Pr, Qr, u, i, j, r1, r2, loss_type=lid)
            Pr, Qr = ref.step([Pr, Qr], [gP, gQ])
            want.append(loss_k)
        np.testing.assert_allclose(sl.cpu().numpy(), want, rtol=2e-5)
        assert abs(float(ctx.epoch_acc[0].cpu()) - sum(want)) <= 2e-5 * abs(sum(want)) and float(ctx.epoch_acc[1].cpu()) == 0
        # (a gradient that cancels to round-off takes Adam's +-lr step with a sign the summation order decides: rare
        # elements up to 2 lr per step apart, tests/test_gpu_fuzz.py; everything else at fp32 round-off of the updates)
        for got, ref_t in ((P, Pr), (Q, Qr)):
            diff = np.abs(got.cpu().numpy() - ref_t)
            assert (diff > 5e-5).mean() < 0.01 and diff.max() <= 2.5 * lr * nb * epoch, (epoch, float(diff.max()))
    assert adam.t == 2 * nb
    assert int(adam.last[0].min().cpu()) == adam.t and int(adam.last[1].min().cpu()) == adam.t      # flushed
    ctx.close(); plan.close()


def test_adam_fit_runs_the_epoch_call():
    """MF.fit with Adam takes the one-enqueue epoch (daisy_bpr_fit_epoch_adam) and equals the fit driven step by step:
    the same epoch losses, the same tables."""
    import logging
    from daisyrec_amd import ops
    from daisyrec_amd.model.MFRecommender import MF
    from daisyrec_amd.utils.dataset import BasicDataset, get_dataloader
    U, I, n, B, d = 300, 120, 5000, 512, 32
    tri = _triples(n, U, I, 77)
    cfg = {"gpu": "0", "logger": logging.getLogger("t"), "lr": 0.01, "reg_1": 0.001, "reg_2": 0.001, "epochs": 2,
           "topk": 10, "user_num": U, "item_num": I, "factors": d, "loss_type": "BPR", "optimizer": "adam",
           "init_method": "default", "early_stop": False, "progress": False, "seed": 7}
    out = []
    for one_call in (True, False):
        torch.manual_seed(5)
        m = MF(cfg)
        calls = []
        orig = ops.LazyAdam.fit_epoch
        if one_call:
            ops.LazyAdam.fit_epoch = lambda self, *a, **k: (calls.append(1), orig(self, *a, **k))[1]
        else:                                  # force the step-by-step loop of round 4
            import importlib
            AR = importlib.import_module("daisyrec_amd.model.AbstractRecommender")
            saved = AR._AdamState.staged_epoch

            def stepwise(self, ctx, plan, reg_1, reg_2, loss_id):
                for k in range(plan.num_batches):
                    ctx.set_batch_from_plan(plan, k)
                    self.step(ctx, self._lazy_args[0], self._lazy_args[1], reg_1, reg_2, loss_id, ops.ITEM_MODES["fused"])
                self.flush()
            AR._AdamState.staged_epoch = stepwise
        try:
            torch.manual_seed(11)
            m.fit(get_dataloader(BasicDataset(tri), batch_size=B, shuffle=True, num_workers=0))
        finally:
            ops.LazyAdam.fit_epoch = orig
            if not one_call:
                AR._AdamState.staged_epoch = saved
        if one_call:
            assert len(calls) == 2
        out.append((list(m.epoch_losses), m.embed_user.weight.data.clone(), m.embed_item.weight.data.clone()))
    assert out[0][0] == out[1][0]
    assert torch.equal(out[0][1], out[1][1]) and torch.equal(out[0][2], out[1][2])


@pytest.mark.parametrize("d,loss", [(64, "BPR"), (32, "TL"), (100, "CL")])
@pytest.mark.parametrize("blocks", ["0", "1", None])
def test_long_edge_chains_in_two_levels(d, loss, blocks, monkeypatch):
    """Hot items (round 5): an item that holds a third of a batch's entries is one segment through hundreds of chunks of
    the item pass - a chain of edge records.  Two levels (k_staged_item_edge_blocks: block sums of 32 chunks, then the
    owner hops block by block) against link by link (DAISY_EDGE_BLOCKS=0) and the automatic choice (None: the index
    counted the hottest item, the plan carries its share), each against the oracle; the two-level sums are bitwise
    reproducible.  Chains that end inside a block, at a block boundary, run through several blocks, and several hot
    items side by side."""
    from daisyrec_amd import ops
    if blocks is not None:
        monkeypatch.setenv("DAISY_EDGE_BLOCKS", blocks)
    monkeypatch.setenv("DAISY_STAGED_MERGE", "0")          # (the three-launch form keeps its chains link by link)
    U, I, n, B = 3000, 400, 90_000, 40_000
    rng = np.random.default_rng(d)
    tri = _triples(n, U, I, d + 11)
    hot = rng.random(n)
    tri[hot < 0.45, 1] = 7                                  # ~45 % of the positives, ~22 % of the entries: ~140 chunks of 128
    tri[(hot > 0.5) & (hot < 0.62), 2] = 7                  # ... and some negatives of the same item
    tri[(hot > 0.7) & (hot < 0.80), 1] = 8                  # a second, shorter chain right behind it (~31 chunks)
    tri[(hot > 0.85) & (hot < 0.95), 2] = 399               # and one at the very end of the entry list
    point = loss in ("CL", "SL")
    if point:
        tri[:, 2] = rng.integers(0, 2, n)
    P0, Q0 = _tables(U, I, d, d + 1, scale=0.05)
    t_dev = torch.from_numpy(tri).to(DEV)
    lid = ops.LOSS_IDS[loss]
    runs = []
    for rep in range(2):
        index, plan = ops.TrainIndex(t_dev, U, I, pointwise=point), ops.EpochPlan(n, U, I)
        plan.build_indexed(index, B, order="feistel", seed=5, epoch=1)
        nb = plan.num_batches
        P, Q = torch.from_numpy(P0).to(DEV), torch.from_numpy(Q0).to(DEV)
        ctx = ops.BprContext(B, d, U, I)
        sl = torch.zeros(nb, dtype=torch.float64, device=DEV)
        ctx.fit_epoch_sgd(plan, P, Q, 0.002, 1e-3, 2e-3, loss_type=lid, item_mode=ops.ITEM_MODES["fused"], step_losses=sl)
        torch.cuda.synchronize()
        runs.append((P.clone(), Q.clone(), sl.clone()))
        if rep == 0:
            Pn, Qn = P0.astype(np.float64), Q0.astype(np.float64)
            for k, (rows, _, _) in enumerate(_plan_batches(plan, nb, B)):
                want, Pn, Qn = O.mf_sgd_step(Pn, Qn, rows[:, 0], rows[:, 1], rows[:, 2], 0.002, 1e-3, 2e-3, loss_type=lid)
                assert abs(float(sl[k].cpu()) - want) <= 1e-5 * abs(want), (k, float(sl[k].cpu()), want)
            # the hot row is a sum of ~10^4 fp32 terms: a few hundred ulps of its largest partial sum
            assert np.abs(P.cpu().numpy() - Pn).max() < 5e-6 and np.abs(Q.cpu().numpy() - Qn).max() < 2e-5
        ctx.close(); plan.close(); index.close()
    assert all(torch.equal(a, b) for a, b in zip(runs[0], runs[1]))
