"""Boundary hardening: id-range validation, the sampler's empty-complement
error, fits with fewer rows than one batch, Adam pinned in every item mode and end to end."""
import numpy as np
import pytest
import torch

from conftest import mf_config
from oracle import bpr_mf_numpy as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _t(x):
    return torch.from_numpy(np.ascontiguousarray(x)).to(DEV)


@pytest.fixture(scope="module")
def ops():
    from daisyrec_amd import ops as o
    return o


@pytest.mark.parametrize("mode", ["chunked", "sorted", "atomic"])
def test_kat_adam_every_item_mode(ops, kat_steps, mode):
    """torch.optim.Adam on both tables (AbstractRecommender.py:54) against the reference's own vectors, with
    the default (chunked) gradient kernels as well as the serial ones."""
    from daisyrec_amd.model.AbstractRecommender import _AdamState
    g, name = kat_steps, "bpr_adam"
    U, I, d, B, ns = (int(x) for x in g[f"{name}/meta"])
    lr, r1, r2 = (float(x) for x in g[f"{name}/hyper"])
    P, Q = _t(g[f"{name}/P0"]), _t(g[f"{name}/Q0"])
    ctx = ops.BprContext(B, d, U, I)
    adam = _AdamState(P, Q, lr)
    for s in range(ns):
        ctx.set_batch(_t(g[f"{name}/u"][s]), _t(g[f"{name}/i"][s]), _t(g[f"{name}/j"][s]))
        adam.step(ctx, P, Q, r1, r2, ops.LOSS_IDS["BPR"], ops.ITEM_MODES[mode])
        ref = float(g[f"{name}/loss"][s])
        assert abs(float(ctx.stats[7].cpu()) - ref) <= 1e-5 * abs(ref)
        np.testing.assert_allclose(P.cpu().numpy(), g[f"{name}/P"][s], rtol=0, atol=2e-6)
        np.testing.assert_allclose(Q.cpu().numpy(), g[f"{name}/Q"][s], rtol=0, atol=2e-6)
    ctx.close()


@pytest.mark.parametrize("mode", ["chunked", "sorted"])
def test_kat_adagrad_rmsprop(ops, mode):
    """optim.Adagrad / optim.RMSprop (AbstractRecommender.py:58,60) against the reference's vectors."""
    import os
    from conftest import GOLDEN
    from daisyrec_amd.model.AbstractRecommender import _AdamState
    g = np.load(os.path.join(GOLDEN, "kat_optimizers.npz"))
    for name in g["names"]:
        U, I, d, B, ns = (int(x) for x in g[f"{name}/meta"])
        lr, r1, r2 = (float(x) for x in g[f"{name}/hyper"])
        P, Q = _t(g[f"{name}/P0"]), _t(g[f"{name}/Q0"])
        ctx = ops.BprContext(B, d, U, I)
        st = _AdamState(P, Q, lr, kind=str(g[f"{name}/optimizer"]))
        for s in range(ns):
            ctx.set_batch(_t(g[f"{name}/u"][s]), _t(g[f"{name}/i"][s]), _t(g[f"{name}/j"][s]))
            st.step(ctx, P, Q, r1, r2, ops.LOSS_IDS[str(g[f"{name}/loss_type"])], ops.ITEM_MODES[mode])
            ref = float(g[f"{name}/loss"][s])
            assert abs(float(ctx.stats[7].cpu()) - ref) <= 1e-5 * abs(ref), (name, s)
            np.testing.assert_allclose(P.cpu().numpy(), g[f"{name}/P"][s], rtol=0, atol=4e-6, err_msg=f"{name} {s}")
            np.testing.assert_allclose(Q.cpu().numpy(), g[f"{name}/Q"][s], rtol=0, atol=4e-6, err_msg=f"{name} {s}")
        ctx.close()


def test_every_mirror_trains_with_adagrad_and_rmsprop():
    """the optimiser name travels through MF / FM fit (dense state per tensor)"""
    from daisyrec_amd.model.FMRecommender import FM
    from daisyrec_amd.model.MFRecommender import MF
    from daisyrec_amd.utils.dataset import BasicDataset, get_dataloader
    rng = np.random.default_rng(4)
    tri = np.stack([rng.integers(0, 30, 600), rng.integers(0, 20, 600), rng.integers(0, 20, 600)], 1).astype(np.int32)
    for cls in (MF, FM):
        for opt in ("adagrad", "rmsprop"):
            model = cls(mf_config(user_num=30, item_num=20, epochs=3, factors=8, optimizer=opt, lr=0.01))
            model.fit(get_dataloader(BasicDataset(tri), batch_size=128, shuffle=True, num_workers=0))
            assert len(model.epoch_losses) == 3 and model.epoch_losses[-1] < model.epoch_losses[0]
        with pytest.raises(RuntimeError, match="SparseAdam"):
            cls(mf_config(user_num=30, item_num=20, epochs=1, factors=8, optimizer="sparse_adam")).fit(
                get_dataloader(BasicDataset(tri), batch_size=128, shuffle=True, num_workers=0))


def test_ml100k_c1_adam_through_the_dropin():
    """BASELINE configs[0] with `--optimizer adam --lr 0.001`: 3 epochs of dense Adam through MF.fit against the
    golden reference run (tests/golden/ml100k_c1_adam.npz, make_golden.py)."""
    import os
    from conftest import GOLDEN
    from daisyrec_amd.model.MFRecommender import MF
    from daisyrec_amd.utils.dataset import BasicDataset, CandidatesDataset, get_dataloader
    g = np.load(os.path.join(GOLDEN, "ml100k_c1_adam.npz"))
    cfg = mf_config(user_num=int(g["user_num"]), item_num=int(g["item_num"]), epochs=int(g["epochs"]),
                    optimizer="adam", lr=0.001)
    torch.manual_seed(int(g["seed"]))
    model = MF(cfg)
    np.testing.assert_array_equal(model.embed_user.weight.detach().numpy(), g["P0"])
    loader = get_dataloader(BasicDataset(g["samples"]), batch_size=int(g["batch_size"]), shuffle=True, num_workers=0)
    torch.set_rng_state(torch.from_numpy(g["rng_state_before_fit"]))
    model.fit(loader)
    for got, ref in zip(model.epoch_losses, g["epoch_losses"]):
        assert abs(got - ref) <= 1e-5 * abs(ref), (got, ref)
    # Adam turns a gradient element that is pure rounding residue into a +-lr step (in both implementations):
    # a handful of elements differ by a few lr, everything else agrees to round-off
    for got, ref in ((model.embed_user.weight, g["P1"]), (model.embed_item.weight, g["Q1"])):
        diff = np.abs(got.detach().cpu().numpy() - ref)
        assert (diff < 5e-4).mean() > 0.999 and diff.max() < 5e-3, (diff.max(), (diff < 5e-4).mean())
    ucands = [[int(u), c] for u, c in zip(g["test_u"], g["cands"])]
    preds = model.rank(get_dataloader(CandidatesDataset(ucands), batch_size=128, shuffle=False, num_workers=0))
    same = (preds == g["preds"]).all(1).mean()
    print("MF-Adam ml-100k: users with an identical top-50 list:", same)
    assert same >= 0.97            # Adam divides by sqrt(v): last-ulp differences of tiny gradients are amplified


def test_out_of_range_ids_raise_like_the_reference(ops):
    """nn.Embedding raises IndexError in the reference (MFRecommender.py:64-65); the native side never
    dereferences such an id and reports it."""
    U, I, d, B = 50, 40, 16, 64
    rng = np.random.default_rng(0)
    u, i, j = (rng.integers(0, n, B).astype(np.int32) for n in (U, I, I))
    ctx = ops.BprContext(B, d, U, I)
    P, Q = torch.zeros(U, d, device=DEV), torch.zeros(I, d, device=DEV)
    for col, bad in ((0, U), (0, -1), (1, I), (2, I + 5), (2, -1)):
        cols = [u.copy(), i.copy(), j.copy()]
        cols[col][7] = bad
        with pytest.raises(ValueError, match="out of range"):
            ctx.set_batch(*(_t(c) for c in cols))
        # unvalidated: ids are neutralised, the step still runs inside the tables
        ctx.set_batch(*(_t(c) for c in cols), validate=False)
        ctx.sgd_step(P, Q, 0.1, 0.0, 0.0, item_mode=ops.ITEM_MODES["chunked"])
        torch.cuda.synchronize()
    ctx.set_batch(_t(u), _t(i), _t(j))                      # a good batch passes
    tri = np.stack([u, i, j], 1).astype(np.int32)
    plan = ops.EpochPlan(B, U, I)
    bad = tri.copy()
    bad[3, 1] = I
    with pytest.raises(ValueError, match="out of range"):
        plan.build(_t(bad), 16)
    plan.build(_t(tri), 16)
    with pytest.raises(ValueError, match="permutation"):
        plan.build(_t(tri), 16, order="perm", perm=_t(np.arange(B, dtype=np.int64) + 1))
    # point-wise rows: the third column is a label, not an id
    ctx.set_pointwise(True)
    lab = np.full(B, 7, np.int32)
    ctx.set_batch(_t(u), _t(i), _t(lab))
    ctx.close(); plan.close()


def test_fit_rejects_out_of_range_triples():
    from daisyrec_amd.model.MFRecommender import MF
    from daisyrec_amd.utils.dataset import BasicDataset, get_dataloader
    rng = np.random.default_rng(1)
    tri = np.stack([rng.integers(0, 30, 500), rng.integers(0, 20, 500), rng.integers(0, 20, 500)], 1).astype(np.int32)
    tri[17, 2] = 20
    for opt in ("sgd", "adam"):
        model = MF(mf_config(user_num=30, item_num=20, epochs=1, factors=8, optimizer=opt))
        with pytest.raises(ValueError, match="out of range"):
            model.fit(get_dataloader(BasicDataset(tri), batch_size=64, shuffle=True, num_workers=0))


def test_sampler_raises_when_a_user_has_no_negatives():
    """sampler.py:84-89: np.random.choice on an empty complement raises ValueError for ANY user id in
    range(user_num) that interacted with every item."""
    import pandas as pd
    from daisyrec_amd.utils.sampler import BasicNegtiveSampler
    users = np.array([0, 0, 0, 1, 2], dtype=np.int32)
    items = np.array([0, 1, 2, 1, 0], dtype=np.int32)        # user 0 has all 3 items
    df = pd.DataFrame({"user": users, "item": items, "rating": 1.0})
    ur = {0: {0, 1, 2}, 1: {1}, 2: {0}}
    cfg = mf_config(user_num=3, item_num=3, num_ng=2, train_ur=ur)
    with pytest.raises(ValueError, match="cannot be empty"):
        BasicNegtiveSampler(df, cfg).sampling()
    cfg = mf_config(user_num=3, item_num=4, num_ng=2, train_ur=ur)
    tri = BasicNegtiveSampler(df, cfg).sampling()
    assert (tri[tri[:, 0] == 0][:, 2] == 3).all()
    # no train_ur: duplicate (user, item) rows of df do not corrupt the complement search
    df2 = pd.concat([df, df.iloc[:2]], ignore_index=True)
    cfg = mf_config(user_num=3, item_num=5, num_ng=8)
    tri = BasicNegtiveSampler(df2, cfg).sampling()
    assert set(tri[tri[:, 0] == 0][:, 2].tolist()) <= {3, 4} and len(tri) == 8 * len(df2)


@pytest.mark.parametrize("opt,loss", [("sgd", "BPR"), ("adam", "BPR"), ("sgd", "CL")])
def test_fit_with_fewer_rows_than_one_batch(opt, loss):
    """len(dataset) < batch_size: the DataLoader serves one partial batch per epoch; so does fit."""
    from daisyrec_amd.model.MFRecommender import MF
    from daisyrec_amd.utils.dataset import BasicDataset, get_dataloader
    rng = np.random.default_rng(2)
    n, U, I, d = 100, 30, 20, 8
    third = rng.integers(0, 2 if loss == "CL" else I, n)
    tri = np.stack([rng.integers(0, U, n), rng.integers(0, I, n), third], 1).astype(np.int32)
    torch.manual_seed(3)
    model = MF(mf_config(user_num=U, item_num=I, epochs=2, factors=d, optimizer=opt, loss_type=loss, batch_size=256))
    P0, Q0 = model.embed_user.weight.detach().numpy().copy(), model.embed_item.weight.detach().numpy().copy()
    model.fit(get_dataloader(BasicDataset(tri), batch_size=256, shuffle=True, num_workers=0))
    assert len(model.epoch_losses) == 2 and all(np.isfinite(model.epoch_losses))
    if opt == "sgd":
        Pn, Qn = P0, Q0
        for _ in range(2):
            want, Pn, Qn = O.mf_sgd_step(Pn, Qn, tri[:, 0], tri[:, 1], tri[:, 2], 0.01, 0.001, 0.001,
                                         O.LOSS_IDS[loss])
        assert abs(model.epoch_losses[-1] - want) <= 1e-5 * abs(want)
        np.testing.assert_allclose(model.embed_user.weight.detach().cpu().numpy(), Pn, atol=3e-6)
    # drop_last with fewer rows than a batch: nothing to train on, like an empty loader
    model = MF(mf_config(user_num=U, item_num=I, epochs=3, factors=d))
    from torch.utils.data import DataLoader
    model.fit(DataLoader(BasicDataset(tri), batch_size=256, shuffle=False, drop_last=True))
    assert model.epoch_losses == [0.0, 0.0, 0.0]


def test_rank_truncates_topk_to_the_candidate_count(ops):
    rng = np.random.default_rng(5)
    P, Q = _t(rng.standard_normal((10, 8)).astype(np.float32)), _t(rng.standard_normal((30, 8)).astype(np.float32))
    us = torch.arange(4, device=DEV)
    cands = _t(rng.integers(0, 30, (4, 6)))
    ids = ops.mf_rank_topk(P, Q, us, cands, 50)
    assert tuple(ids.shape) == (4, 6)                       # rank_list[:, :topk] on 6 candidates
    want, _ = O.mf_rank(P.cpu().numpy(), Q.cpu().numpy(), us.cpu().numpy(), cands.cpu().numpy(), 6)
    np.testing.assert_array_equal(ids.cpu().numpy().astype(np.float32), want)
    assert ops.mf_full_rank(P, Q, 2, 100).numel() == 30


def test_lazy_adam_equals_dense_adam_bit_for_bit():
    """ops.LazyAdam (rows without a gradient replayed in registers when next needed) against the dense optimiser
    that rewrites every row in every step: tables and both moments identical after every flush - batches that touch
    a small part of large tables, rows touched in consecutive steps, rows never touched, d that does not fill its
    lanes, more steps than the table was sized for."""
    from daisyrec_amd import ops
    from daisyrec_amd.model.AbstractRecommender import _AdamState
    rng = np.random.default_rng(8)
    for d, U, I, B, steps in ((64, 5000, 3000, 300, 25), (100, 400, 300, 700, 12), (20, 50, 40, 64, 40)):
        P0 = (rng.standard_normal((U, d)) * 0.1).astype(np.float32)
        Q0 = (rng.standard_normal((I, d)) * 0.1).astype(np.float32)
        batches = [tuple(torch.from_numpy(rng.integers(0, hi, B).astype(np.int32)).to("cuda") for hi in (U, I, I))
                   for _ in range(steps)]
        res = []
        for lazy in (False, True):
            P, Q = torch.from_numpy(P0).to("cuda"), torch.from_numpy(Q0).to("cuda")
            ctx = ops.BprContext(B, d, U, I)
            st = _AdamState(P, Q, 0.01, None, kind="adam", max_steps=5, lazy=lazy)      # 5: the table must grow
            snaps = []
            for k, (u, i, j) in enumerate(batches):
                ctx.set_batch(u, i, j)
                st.step(ctx, P, Q, 1e-3, 2e-3, 0, ops.ITEM_MODES["chunked"])
                if k % 7 == 6 or k == steps - 1:
                    st.flush()
                    torch.cuda.synchronize()
                    snaps.append((P.clone(), Q.clone()))
            if lazy:
                mom = (st.lazy.m[0], st.lazy.v[0], st.lazy.m[1], st.lazy.v[1])
                assert int(st.lazy.last[0].min()) == steps and int(st.lazy.last[1].min()) == steps
            else:
                sP, sQ = st.opt._state[P.view(-1).data_ptr()], st.opt._state[Q.view(-1).data_ptr()]
                mom = (sP[0].view(U, d), sP[1].view(U, d), sQ[0].view(I, d), sQ[1].view(I, d))
            res.append((snaps, [m.clone() for m in mom]))
            ctx.close()
        for (Pa, Qa), (Pb, Qb) in zip(res[0][0], res[1][0]):
            assert torch.equal(Pa, Pb) and torch.equal(Qa, Qb), (d, U, I, B)
        for a, b in zip(res[0][1], res[1][1]):
            assert torch.equal(a, b), (d, U, I, B)


@pytest.mark.parametrize("path", ["small", "chunked", "staged"])
def test_epoch_stops_at_the_first_non_finite_loss(ops, path):
    """AbstractRecommender.py:122-123: the reference raises when a batch's loss is NaN/inf, BEFORE that batch's
    backward.  The native epoch loop has one host sync per epoch; a device-side flag makes every later kernel of the
    epoch a no-op instead.  Batch k_bad is the first that references an item whose row is NaN: the small-batch and the
    phase loops leave both tables exactly as k_bad steps left them; the staged step (forward fused into the user
    update) has moved the user rows of the offending batch, nothing else."""
    rng = np.random.default_rng(3)
    U, I, d = 400, 120, 64
    B = {"small": 128, "chunked": 512, "staged": 512}[path]
    nb, k_bad = 6, 3
    n = nb * B
    u = np.sort(rng.integers(0, U, n)).astype(np.int32)          # identity order: batch k = rows [k*B, (k+1)*B)
    i = rng.integers(0, 100, n).astype(np.int32)
    j = rng.integers(0, 100, n).astype(np.int32)
    i[k_bad * B + 5] = 110                                        # the poisoned item enters in batch k_bad
    P0 = (rng.standard_normal((U, d)) * 0.1).astype(np.float32)
    Q0 = (rng.standard_normal((I, d)) * 0.1).astype(np.float32)
    Q0[110] = np.nan
    tri = np.stack([u, i, j], 1)
    mode = ops.ITEM_MODES["fused" if path != "chunked" else "chunked"]

    def run(rows):
        P, Q = _t(P0), _t(Q0)
        t = _t(tri[:rows])
        ctx = ops.BprContext(B, d, U, I)
        plan = ops.EpochPlan(rows, U, I)
        if path == "staged":
            idx = ops.TrainIndex(t, U, I, user_sorted=True)
            plan.build_indexed(idx, B, order="identity")
        else:
            plan.build(t, B, order="identity", user_sorted=True)
        ctx.fit_epoch_sgd(plan, P, Q, 0.05, 1e-3, 1e-3, item_mode=mode)
        acc = ctx.epoch_acc.cpu().numpy().copy()
        out = P.cpu().numpy(), Q.cpu().numpy(), acc
        ctx.close(); plan.close()
        return out

    P_ok, Q_ok, acc_ok = run(k_bad * B)                            # the first k_bad batches alone: all finite
    assert acc_ok[1] == 0
    P_bad, Q_bad, acc_bad = run(n)                                 # the whole epoch
    assert acc_bad[1] == 1, acc_bad                                # exactly ONE non-finite step was counted: the epoch stopped there
    assert np.array_equal(Q_bad[:110], Q_ok[:110]) and np.array_equal(Q_bad[111:], Q_ok[111:])
    if path == "staged":
        touched = np.unique(u[k_bad * B:(k_bad + 1) * B])
        keep = np.setdiff1d(np.arange(U), touched)
        assert np.array_equal(P_bad[keep], P_ok[keep])
    else:
        assert np.array_equal(P_bad, P_ok)


@pytest.mark.parametrize("mode", ["fused", "chunked"])
def test_set_batch_from_triples_selects_rows_anywhere_in_the_array(ops, mode):
    """daisy_bpr_set_batch_from_triples(idx=...): a batch of B rows SELECTED from a larger triple array - what one rank's
    share of a global batch is in the dense-optimiser protocol (sharding.py).  The range check of the selection compared
    its entries against B instead of the array's length until round 4, so any row >= B was refused."""
    rng = np.random.default_rng(17)
    U, I, d, n, B = 300, 200, 64, 5000, 257
    tri = np.stack([rng.integers(0, U, n), rng.integers(0, I, n), rng.integers(0, I, n)], 1).astype(np.int32)
    idx = rng.choice(n, B, replace=False).astype(np.int64)
    idx[0], idx[1] = n - 1, 0                                      # both ends of the array
    P0 = (rng.standard_normal((U, d)) * 0.1).astype(np.float32)
    Q0 = (rng.standard_normal((I, d)) * 0.1).astype(np.float32)
    rows = tri[idx]
    want, Pn, Qn = O.mf_sgd_step(P0, Q0, rows[:, 0], rows[:, 1], rows[:, 2], 0.01, 1e-3, 1e-3)
    P, Q = _t(P0), _t(Q0)
    ctx = ops.BprContext(B, d, U, I)
    sl = torch.zeros(1, dtype=torch.float64, device="cuda")
    ctx.set_batch_from_triples(_t(tri), idx=_t(idx))
    ctx.sgd_step(P, Q, 0.01, 1e-3, 1e-3, item_mode=ops.ITEM_MODES[mode], step_loss=sl)
    torch.cuda.synchronize()
    assert abs(float(sl.cpu()) - want) <= 1e-5 * abs(want)
    assert np.abs(P.cpu().numpy() - Pn).max() < 3e-6 and np.abs(Q.cpu().numpy() - Qn).max() < 3e-6
    bad = idx.copy()
    bad[5] = n                                                     # one past the array
    with pytest.raises(ValueError, match="out of range"):
        ctx.set_batch_from_triples(_t(tri), idx=_t(bad))
    ctx.close()


@pytest.mark.parametrize("B", [512, 4096])
def test_bare_step_with_a_non_finite_loss_is_the_same_in_both_launch_forms(ops, B, monkeypatch):
    """a step WITHOUT an epoch accumulator has no halt word; the four-launch form then applies the whole step
    whatever its loss, and the three-launch form (batches up to 16 384 samples) has to do the same - its item workgroups
    used to stop themselves on a non-finite loss while the edge launches, which have no word to look at, still committed:
    a half-updated Q that depended on the batch size."""
    rng = np.random.default_rng(8)
    U, I, d = 300, 90, 64
    u = np.sort(rng.integers(0, U, B)).astype(np.int32)
    i = rng.integers(0, I - 1, B).astype(np.int32)
    j = rng.integers(0, I - 1, B).astype(np.int32)
    i[7] = I - 1                                              # one sample meets the poisoned row
    P0 = (rng.standard_normal((U, d)) * 0.1).astype(np.float32)
    Q0 = (rng.standard_normal((I, d)) * 0.1).astype(np.float32)
    Q0[I - 1] = np.inf
    out = []
    for merge in ("0", "1"):
        monkeypatch.setenv("DAISY_STAGED_MERGE", merge)
        P, Q = _t(P0), _t(Q0)
        ctx = ops.BprContext(B, d, U, I)
        ctx.set_batch(_t(u), _t(i), _t(j))
        sl = torch.zeros(1, dtype=torch.float64, device="cuda")
        ctx.sgd_step(P, Q, 0.05, 1e-3, 1e-3, item_mode=ops.ITEM_MODES["fused"], step_loss=sl, accumulate=False)
        torch.cuda.synchronize()
        assert not np.isfinite(float(sl.cpu()))
        out.append((P.cpu().numpy(), Q.cpu().numpy()))
        ctx.close()
    # the same rows written in both forms, to summation order (the two forms cut the item pass into chunks of different
    # sizes), NaN / inf patterns included
    for a, b in zip(out[0], out[1]):
        assert np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(np.isinf(a), np.isinf(b))
        assert np.allclose(a, b, rtol=1e-5, atol=1e-7, equal_nan=True)
    assert np.array_equal((out[0][1] != Q0).any(axis=1), (out[1][1] != Q0).any(axis=1))
    moved = (out[0][1][: I - 1] != Q0[: I - 1]).any(axis=1)
    assert moved.sum() > (I - 1) // 2                           # the step WAS applied (no halt word: nothing may stop it)
