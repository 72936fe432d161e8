"""Parity tests proper (`-m gpu`): the HIP path, called through the C ABI, against
(a) the golden vectors produced by the real reference and (b) the CPU oracle on seeded
inputs.  Tolerances: integer/index work bit-exact; fp32 tables within fp32 round-off of
the reference; loss within 1e-5 relative (north_star); ranked top-N identical."""
import os

import numpy as np
import pytest
import torch

from conftest import mf_config
from oracle import bpr_mf_numpy as O

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _t(x, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(x))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(DEV)


@pytest.fixture(scope="module")
def ops():
    from daisyrec_amd import ops as _ops
    return _ops


# ------------------------------------------------------------------ per-step KATs
@pytest.mark.parametrize("item_mode", ["sorted", "atomic", "chunked", "fused"])
def test_kat_steps_sgd(ops, kat_steps, item_mode):
    g = kat_steps
    for name in g["names"]:
        name = str(name)
        if str(g[f"{name}/optimizer"]) != "sgd":
            continue
        U, I, d, B, ns = (int(x) for x in g[f"{name}/meta"])
        lr, r1, r2 = (float(x) for x in g[f"{name}/hyper"])
        lt = ops.loss_id(str(g[f"{name}/loss_type"]))
        P, Q = _t(g[f"{name}/P0"]), _t(g[f"{name}/Q0"])
        ctx = ops.BprContext(B, d, U, I)
        ctx.set_pointwise(lt in ops.POINTWISE_LOSSES)          # CL / SL rows are (user, item, label)
        step_loss = torch.zeros(1, dtype=torch.float64, device=DEV)
        scale = max(1.0, float(np.abs(g[f"{name}/P0"]).max()))
        for s in range(ns):
            ctx.set_batch(_t(g[f"{name}/u"][s]), _t(g[f"{name}/i"][s]), _t(g[f"{name}/j"][s]))
            ctx.sgd_step(P, Q, lr, r1, r2, loss_type=lt, item_mode=ops.ITEM_MODES[item_mode],
                         step_loss=step_loss)
            ref = float(g[f"{name}/loss"][s])
            assert abs(float(step_loss.cpu()) - ref) <= 1e-5 * abs(ref), (name, s)
            np.testing.assert_allclose(P.cpu().numpy(), g[f"{name}/P"][s], rtol=0, atol=2e-6 * scale,
                                       err_msg=f"{name} step {s} P")
            np.testing.assert_allclose(Q.cpu().numpy(), g[f"{name}/Q"][s], rtol=0, atol=2e-6 * scale,
                                       err_msg=f"{name} step {s} Q")
        assert float(ctx.gQ.abs().max().cpu()) == 0.0      # side buffer left clean
        assert abs(float(ctx.epoch_acc[0].cpu()) - float(g[f"{name}/loss"].sum())) <= 1e-5 * float(g[f"{name}/loss"].sum())
        ctx.close()


@pytest.mark.parametrize("item_mode", ["sorted", "fused"])
def test_kat_adam(ops, kat_steps, item_mode):
    """torch.optim.Adam against the reference's vectors: the phase kernels + the dense optimiser ('sorted'), and the
    staged step, whose row owners apply Adam themselves ('fused'; flush() replays the rows without a gradient)"""
    from daisyrec_amd.model.AbstractRecommender import _AdamState
    g, name = kat_steps, "bpr_adam"
    U, I, d, B, ns = (int(x) for x in g[f"{name}/meta"])
    lr, r1, r2 = (float(x) for x in g[f"{name}/hyper"])
    P, Q = _t(g[f"{name}/P0"]), _t(g[f"{name}/Q0"])
    ctx = ops.BprContext(B, d, U, I)
    adam = _AdamState(P, Q, lr)
    for s in range(ns):
        ctx.set_batch(_t(g[f"{name}/u"][s]), _t(g[f"{name}/i"][s]), _t(g[f"{name}/j"][s]))
        adam.step(ctx, P, Q, r1, r2, ops.LOSS_IDS["BPR"], ops.ITEM_MODES[item_mode])
        adam.flush()
        ref = float(g[f"{name}/loss"][s])
        assert abs(float(ctx.stats[7].cpu()) - ref) <= 1e-5 * abs(ref)
        np.testing.assert_allclose(P.cpu().numpy(), g[f"{name}/P"][s], rtol=0, atol=2e-6)
        np.testing.assert_allclose(Q.cpu().numpy(), g[f"{name}/Q"][s], rtol=0, atol=2e-6)
    ctx.close()


# ------------------------------------------------------------------ oracle sweeps
@pytest.mark.parametrize("d", [1, 3, 8, 16, 20, 32, 48, 64, 100, 128, 200, 256, 300, 512])
def test_step_vs_oracle_shapes(ops, d):
    """Every fragment shape of dispatch_d (vector and scalar paths, ragged tails)."""
    rng = np.random.default_rng(d)
    U, I, B = 37, 53, 77
    P0 = (rng.standard_normal((U, d)) * 0.2).astype(np.float32)
    Q0 = (rng.standard_normal((I, d)) * 0.2).astype(np.float32)
    u = rng.integers(0, U, B).astype(np.int32)
    i = rng.integers(0, I, B).astype(np.int32)
    j = rng.integers(0, I, B).astype(np.int32)
    lr, r1, r2 = 0.05, 0.01, 0.02
    lab = rng.integers(0, 2, B).astype(np.int32)
    for lt_name in ("BPR", "HL", "TL", "CL", "SL"):
        third = lab if lt_name in ("CL", "SL") else j
        loss, Pn, Qn = O.mf_sgd_step(P0, Q0, u, i, third, lr, r1, r2, O.LOSS_IDS[lt_name])
        for mode in ("sorted", "atomic", "chunked", "fused"):
            P, Q = _t(P0), _t(Q0)
            ctx = ops.BprContext(B, d, U, I)
            ctx.set_pointwise(lt_name in ("CL", "SL"))
            sl = torch.zeros(1, dtype=torch.float64, device=DEV)
            ctx.set_batch(_t(u), _t(i), _t(third))
            ctx.sgd_step(P, Q, lr, r1, r2, loss_type=ops.LOSS_IDS[lt_name],
                         item_mode=ops.ITEM_MODES[mode], step_loss=sl)
            assert abs(float(sl.cpu()) - loss) <= 1e-5 * abs(loss), (d, lt_name, mode)
            np.testing.assert_allclose(P.cpu().numpy(), Pn, rtol=0, atol=3e-6, err_msg=f"{d} {lt_name} {mode}")
            np.testing.assert_allclose(Q.cpu().numpy(), Qn, rtol=0, atol=3e-6, err_msg=f"{d} {lt_name} {mode}")
            ctx.close()


@pytest.mark.parametrize("B", [1, 2, 63, 64, 65, 255, 256, 257, 1000, 4097])
def test_step_vs_oracle_batch_sizes(ops, B):
    rng = np.random.default_rng(B)
    U, I, d = 200, 150, 64
    P0 = (rng.standard_normal((U, d)) * 0.1).astype(np.float32)
    Q0 = (rng.standard_normal((I, d)) * 0.1).astype(np.float32)
    u = rng.integers(0, U, B).astype(np.int32)
    i = rng.integers(0, I, B).astype(np.int32)
    j = rng.integers(0, I, B).astype(np.int32)
    loss, Pn, Qn = O.mf_sgd_step(P0, Q0, u, i, j, 0.01, 1e-3, 1e-3)
    for mode in ("sorted", "atomic", "chunked", "fused"):
        P, Q = _t(P0), _t(Q0)
        ctx = ops.BprContext(max(B, 8), d, U, I)
        sl = torch.zeros(1, dtype=torch.float64, device=DEV)
        ctx.set_batch(_t(u), _t(i), _t(j))
        ctx.sgd_step(P, Q, 0.01, 1e-3, 1e-3, item_mode=ops.ITEM_MODES[mode], step_loss=sl)
        assert abs(float(sl.cpu()) - loss) <= 1e-5 * abs(loss)
        np.testing.assert_allclose(P.cpu().numpy(), Pn, rtol=0, atol=3e-6)
        np.testing.assert_allclose(Q.cpu().numpy(), Qn, rtol=0, atol=3e-6)
        ctx.close()


def test_sorted_mode_is_bitwise_reproducible(ops):
    rng = np.random.default_rng(5)
    U, I, d, B = 500, 300, 64, 5000
    P0 = (rng.standard_normal((U, d)) * 0.1).astype(np.float32)
    Q0 = (rng.standard_normal((I, d)) * 0.1).astype(np.float32)
    u, i, j = (rng.integers(0, n, B).astype(np.int32) for n in (U, I, I))
    outs = []
    for _ in range(3):
        P, Q = _t(P0), _t(Q0)
        ctx = ops.BprContext(B, d, U, I)
        for _s in range(3):
            ctx.set_batch(_t(u), _t(i), _t(j))
            ctx.sgd_step(P, Q, 0.01, 1e-3, 1e-3, item_mode=ops.ITEM_MODES["sorted"])
        outs.append((P.cpu().numpy().copy(), Q.cpu().numpy().copy(), float(ctx.epoch_acc[0].cpu())))
        ctx.close()
    for o in outs[1:]:
        assert np.array_equal(o[0], outs[0][0]) and np.array_equal(o[1], outs[0][1]) and o[2] == outs[0][2]


def test_zero_lr_is_identity_and_errors(ops):
    rng = np.random.default_rng(1)
    U, I, d, B = 64, 64, 64, 512
    P0 = (rng.standard_normal((U, d)) * 0.1).astype(np.float32)
    Q0 = (rng.standard_normal((I, d)) * 0.1).astype(np.float32)
    u, i, j = (rng.integers(0, n, B).astype(np.int32) for n in (U, I, I))
    P, Q = _t(P0), _t(Q0)
    ctx = ops.BprContext(B, d, U, I)
    with pytest.raises(RuntimeError):                      # call-order violation
        ctx.forward(P, Q)
    ctx.set_batch(_t(u), _t(i), _t(j))
    ctx.sgd_step(P, Q, 0.0, 1e-3, 1e-3)
    assert np.array_equal(P.cpu().numpy(), P0) and np.array_equal(Q.cpu().numpy(), Q0)
    with pytest.raises(NotImplementedError):               # MFRecommender.py:90-91
        ctx.forward(P, Q, loss_type=7)
    with pytest.raises(ValueError):                        # batch larger than the context
        big = _t(np.zeros(B + 1, dtype=np.int32))
        ctx.set_batch(big, big, big)
    ctx.close()


# ------------------------------------------------------------------ ranking
def test_rank_kat(ops, rank_kat):
    r = rank_kat
    P, Q = _t(r["P"]), _t(r["Q"])
    topk = int(r["topk"])
    ids, scores = ops.mf_rank_topk(P, Q, _t(r["us"]), _t(r["cands"]), topk, return_scores=True)
    np.testing.assert_array_equal(ids.cpu().numpy().astype(np.float32), r["preds"])
    _, ref_scores = O.mf_rank(r["P"], r["Q"], r["us"], r["cands"], topk)
    np.testing.assert_allclose(scores.cpu().numpy(), ref_scores, rtol=1e-5, atol=1e-7)
    full = np.stack([ops.mf_full_rank(P, Q, int(u), topk).cpu().numpy() for u in r["us"]])
    np.testing.assert_array_equal(full, r["full"])
    pred = ops.mf_predict(P, Q, _t(r["us"]), _t(r["cands"][:, 0]))
    np.testing.assert_allclose(pred.cpu().numpy(), r["predict"], rtol=1e-5, atol=1e-7)


def test_rank_ties_and_shapes(ops):
    """Duplicate candidates give exact score ties: order must be by candidate position
    (stable), as torch.argsort(descending=True) on the reference path."""
    rng = np.random.default_rng(3)
    U, I, d = 9, 40, 20
    P = (rng.standard_normal((U, d))).astype(np.float32)
    Q = (rng.standard_normal((I, d))).astype(np.float32)
    us = np.arange(U, dtype=np.int64)
    cands = rng.integers(0, 6, size=(U, 33)).astype(np.int64)       # heavy duplication
    want, _ = O.mf_rank(P, Q, us, cands, 33)
    got = ops.mf_rank_topk(_t(P), _t(Q), _t(us), _t(cands), 33).cpu().numpy()
    np.testing.assert_array_equal(got.astype(np.float32), want)
    one = ops.mf_rank_topk(_t(P), _t(Q), _t(us[:1]), _t(cands[:1]), 5).cpu().numpy()   # 1-row batch
    np.testing.assert_array_equal(one.astype(np.float32), want[:1, :5])


# ------------------------------------------------------------------ sampler / loader (integer work: bit exact)
def _csr(users, items, U):
    order = np.lexsort((items, users))
    indptr = np.zeros(U + 1, dtype=np.int64)
    np.add.at(indptr, users.astype(np.int64) + 1, 1)
    return np.cumsum(indptr), items[order]


def test_sampler_bit_exact_vs_oracle(ops, ml100k):
    g = ml100k
    U, I = int(g["user_num"]), int(g["item_num"])
    users, items = g["train_users"], g["train_items"]
    indptr_o, csr_o = _csr(users, items, U)
    indptr, csr = ops.build_user_csr(_t(users), _t(items), U)
    np.testing.assert_array_equal(indptr.cpu().numpy(), indptr_o)
    np.testing.assert_array_equal(csr.cpu().numpy(), csr_o)
    for num_ng, epoch in ((1, 0), (4, 3)):
        js = ops.sample_neg_per_user(indptr, csr, I, num_ng, 2022, epoch).cpu().numpy()
        np.testing.assert_array_equal(js, O.sample_uniform_neg_per_user(indptr_o, csr_o, I, num_ng, 2022, epoch))
        tri = ops.expand_triples(_t(users), _t(items), _t(js)).cpu().numpy()
        np.testing.assert_array_equal(tri, O.expand_triples(users, items, js))
        pos = {(int(a), int(b)) for a, b in zip(users, items)}
        assert not any((int(a), int(c)) in pos for a, c in zip(tri[:, 0], tri[:, 2]))
    tri_d = _t(O.expand_triples(users, items, np.zeros((U, 1), np.int32))[:5000])
    ops.resample_neg_per_interaction(indptr, csr, I, tri_d, 11, 2)
    want = O.sample_uniform_neg_per_interaction(indptr_o, csr_o, users[:5000], I, 1, 11, 2)
    np.testing.assert_array_equal(tri_d.cpu().numpy()[:, 2], want[:, 0])


def test_sampler_edge_cases(ops):
    # user 0: no positives; user 1: every item positive (-1); user 2: all but one
    users = np.array([1, 1, 1, 1, 2, 2, 2], dtype=np.int32)
    items = np.array([3, 0, 2, 1, 0, 1, 3], dtype=np.int32)
    indptr, csr = ops.build_user_csr(_t(users), _t(items), 3)
    js = ops.sample_neg_per_user(indptr, csr, 4, 64, 1, 0).cpu().numpy()
    assert set(js[0].tolist()) == {0, 1, 2, 3}
    assert (js[1] == -1).all() and (js[2] == 2).all()


def test_randperm_is_the_oracle_permutation(ops):
    n = 100003
    perm = ops.randperm(n, 2022, 5).cpu().numpy()
    keys = np.array([O._draw_u64(2022, 5 | (1 << 62), e) for e in range(2000)], dtype=np.uint64)
    assert np.array_equal(np.sort(perm), np.arange(n))
    sub = perm[np.isin(perm, np.arange(2000))]
    np.testing.assert_array_equal(sub, np.argsort(keys, kind="stable"))
    assert not np.array_equal(perm, ops.randperm(n, 2022, 6).cpu().numpy())


def test_sampler_mirror_interface(ops, ml100k):
    """BasicNegtiveSampler(df, config).sampling() (sampler.py:14-103) through the HIP kernels."""
    import pandas as pd
    from daisyrec_amd.utils.sampler import BasicNegtiveSampler
    g = ml100k
    df = pd.DataFrame({"user": g["train_users"], "item": g["train_items"], "rating": 1.0})
    cfg = mf_config(user_num=int(g["user_num"]), item_num=int(g["item_num"]), num_ng=2)
    tri = BasicNegtiveSampler(df, cfg).sampling()
    assert tri.dtype == np.int32 and tri.shape == (2 * len(df), 3)
    np.testing.assert_array_equal(tri[::2, 0], g["train_users"])
    np.testing.assert_array_equal(tri[::2, 1], g["train_items"])
    first = {}
    for row in tri.reshape(-1, 2, 3):
        key = int(row[0, 0])
        val = (int(row[0, 2]), int(row[1, 2]))
        assert first.setdefault(key, val) == val            # same negatives for every row of a user
    ur = {}
    for u, i in zip(g["train_users"], g["train_items"]):
        ur.setdefault(int(u), set()).add(int(i))
    assert all(int(j) not in ur[int(u)] for u, _, j in tri)
    cfg2 = dict(cfg, train_ur=ur)
    np.testing.assert_array_equal(BasicNegtiveSampler(df, cfg2).sampling(), tri)


def test_popularity_mixed_sampling(ops, ml100k):
    """sampler.py:44-54,65-81 ('high-pop' / 'low-pop' with sample_ratio > 0): int(ratio * num_ng) of a user's
    negatives from the popularity distribution over all items, the rest uniform from the complement; the device
    draws bit-exact against the oracle's inverse-CDF restatement, the empirical distribution against pop_prob."""
    import pandas as pd
    from daisyrec_amd.utils.sampler import BasicNegtiveSampler
    g = ml100k
    U, I = int(g["user_num"]), int(g["item_num"])
    df = pd.DataFrame({"user": g["train_users"], "item": g["train_items"], "rating": 1})
    ur = {}
    for u, i in zip(g["train_users"], g["train_items"]):
        ur.setdefault(int(u), set()).add(int(i))
    for u in range(U):
        ur.setdefault(u, set())
    for method in ("high-pop", "low-pop"):
        cfg = mf_config(user_num=U, item_num=I, num_ng=5, sample_method=method, sample_ratio=0.5, train_ur=ur)
        smp = BasicNegtiveSampler(df, cfg)
        tri = smp.sampling()
        assert tri.shape == (len(df) * 5, 3) and tri.dtype == np.int32
        js = tri[:, 2].reshape(-1, 5)
        users = g["train_users"]
        # columns 0..2 uniform from the complement (never a positive), columns 3..4 from the popularity distribution
        assert all(int(j) not in ur[int(u)] for u, row in zip(users[::37], js[::37]) for j in row[:3])
        cdf = np.cumsum(smp.pop_prob.astype(np.float64))
        want = O.sample_categorical(cdf, 40, 2, int(cfg["seed"]), (1 << 62) | 0)
        first_row_of_user = {int(u): k for k, u in reversed(list(enumerate(users)))}
        for u in range(40):
            if u in first_row_of_user:
                np.testing.assert_array_equal(js[first_row_of_user[u], 3:], want[u])
        # distribution of many draws: chi-square against pop_prob on the items with a non-negligible expectation
        draws = ops.sample_categorical(torch.from_numpy(cdf).to(DEV), 4000, 50, 11, 5).cpu().numpy().ravel()
        cnt = np.bincount(draws, minlength=I).astype(np.float64)
        exp = smp.pop_prob * draws.size
        keep = exp >= 5
        chi2 = ((cnt[keep] - exp[keep]) ** 2 / exp[keep]).sum()
        dof = int(keep.sum()) - 1
        assert abs(chi2 - dof) < 6 * np.sqrt(2 * dof), (method, chi2, dof)
        if method == "high-pop":
            assert cnt[smp.pop_prob == 0].sum() == 0         # items nobody interacted with are never drawn
    cfg = mf_config(user_num=U, item_num=I, num_ng=4, sample_method="high-pop", sample_ratio=1.0, train_ur=ur)
    assert BasicNegtiveSampler(df, cfg).sampling().shape == (len(df) * 4, 3)      # no uniform share at all


# ------------------------------------------------------------------ BASELINE config C1 end to end
def test_ml100k_c1_through_the_dropin(ml100k):
    """ml-100k, d=32, num_ng=1, SGD, B=256 (BASELINE.json configs[0]) through MF.fit/MF.rank with
    the reference's own triples, init and DataLoader order: epoch losses within 1e-5
    (relative) of the reference CPU run, ranked top-N identical."""
    from daisyrec_amd.model.MFRecommender import MF
    from daisyrec_amd.utils.dataset import BasicDataset, CandidatesDataset, get_dataloader
    g = ml100k
    cfg = mf_config(user_num=int(g["user_num"]), item_num=int(g["item_num"]), epochs=int(g["epochs"]))
    torch.manual_seed(int(g["seed"]))
    model = MF(cfg)
    np.testing.assert_array_equal(model.embed_user.weight.detach().numpy(), g["P0"])
    loader = get_dataloader(BasicDataset(g["samples"]), batch_size=int(g["batch_size"]), shuffle=True,
                            num_workers=4)
    torch.set_rng_state(torch.from_numpy(g["rng_state_before_fit"]))
    model.fit(loader)
    assert len(model.epoch_losses) == int(g["epochs"])
    for got, ref in zip(model.epoch_losses, g["epoch_losses"]):
        assert abs(got - ref) <= 1e-5 * abs(ref), (got, ref)
    np.testing.assert_allclose(model.embed_user.weight.detach().cpu().numpy(), g["P1"], atol=2e-4)
    np.testing.assert_allclose(model.embed_item.weight.detach().cpu().numpy(), g["Q1"], atol=2e-4)
    ucands = [[int(u), c] for u, c in zip(g["test_u"], g["cands"])]
    test_loader = get_dataloader(CandidatesDataset(ucands), batch_size=128, shuffle=False, num_workers=0)
    preds = model.rank(test_loader)
    assert preds.dtype == np.float32 and preds.shape == g["preds"].shape
    np.testing.assert_array_equal(preds, g["preds"])
    full = np.stack([model.full_rank(int(u)) for u in g["test_u"][:16]])
    np.testing.assert_array_equal(full, g["full_rank16"])
    assert abs(model.predict(3, 5) - float(O.mf_forward(g["P1"], g["Q1"], [3], [5])[0])) < 1e-4
    # calc_loss on one collated batch (MFRecommender.py:70-97)
    b = g["samples"][:256]
    want = O.mf_pair_grad(model.embed_user.weight.detach().cpu().numpy(),
                          model.embed_item.weight.detach().cpu().numpy(), b[:, 0], b[:, 1], b[:, 2],
                          cfg["reg_1"], cfg["reg_2"])[0]
    got = float(model.calc_loss([torch.from_numpy(b[:, k].copy()) for k in range(3)]).cpu())
    assert abs(got - want) <= 1e-5 * abs(want)


def test_ml100k_default_run_through_the_dropin():
    """What `python run_examples/test.py` trains when nothing is overridden: ml-100k, factors = 100 (mf.yaml:1),
    num_ng = 4 and batch_size = 256 (basic.yaml:22-24), SGD, BPR - 313 452 triples, 1225 steps per epoch, rows that
    do not fill their lanes.  Fixture generated from the reference by tests/golden/make_golden.py."""
    from daisyrec_amd.model.MFRecommender import MF
    from daisyrec_amd.utils.dataset import BasicDataset, CandidatesDataset, get_dataloader
    from conftest import GOLDEN
    g = np.load(os.path.join(GOLDEN, "ml100k_default.npz"))
    assert int(g["factors"]) == 100 and int(g["batch_size"]) == 256 and g["samples"].shape[0] == 313452
    cfg = mf_config(user_num=int(g["user_num"]), item_num=int(g["item_num"]), epochs=int(g["epochs"]),
                    factors=100, num_ng=4)
    torch.manual_seed(int(g["seed"]))
    model = MF(cfg)
    np.testing.assert_array_equal(model.embed_user.weight.detach().numpy(), g["P0"])
    loader = get_dataloader(BasicDataset(g["samples"]), batch_size=256, shuffle=True, num_workers=0)
    torch.set_rng_state(torch.from_numpy(g["rng_state_before_fit"]))
    model.fit(loader)
    for got, ref in zip(model.epoch_losses, g["epoch_losses"]):
        assert abs(got - ref) <= 1e-5 * abs(ref), (got, ref)
    np.testing.assert_allclose(model.embed_user.weight.detach().cpu().numpy(), g["P1"], atol=5e-4)
    np.testing.assert_allclose(model.embed_item.weight.detach().cpu().numpy(), g["Q1"], atol=5e-4)
    ucands = [[int(u), c] for u, c in zip(g["test_u"], g["cands"])]
    preds = model.rank(get_dataloader(CandidatesDataset(ucands), batch_size=128, shuffle=False, num_workers=0))
    same = (preds == g["preds"]).all(1).mean()
    assert same >= 0.97, same          # 2450 steps amplify summation-order differences; top-50 lists of >= 97 % of the users identical


def test_ml100k_atomic_mode_and_adam_run(ml100k):
    """Throughput mode (fp32 atomics) stays within the same loss tolerance on C1; Adam path trains."""
    from daisyrec_amd.model.MFRecommender import MF
    from daisyrec_amd.utils.dataset import BasicDataset, get_dataloader
    g = ml100k
    cfg = mf_config(user_num=int(g["user_num"]), item_num=int(g["item_num"]), epochs=2, item_mode="fused")
    torch.manual_seed(int(g["seed"]))
    model = MF(cfg)
    loader = get_dataloader(BasicDataset(g["samples"]), batch_size=256, shuffle=True, num_workers=0)
    torch.set_rng_state(torch.from_numpy(g["rng_state_before_fit"]))
    model.fit(loader)
    for got, ref in zip(model.epoch_losses, g["epoch_losses"]):
        assert abs(got - ref) <= 1e-5 * abs(ref)
    cfg = mf_config(user_num=int(g["user_num"]), item_num=int(g["item_num"]), epochs=1, optimizer="adam",
                    batch_size=4096)
    model = MF(cfg)
    loader = get_dataloader(BasicDataset(g["samples"]), batch_size=4096, shuffle=True, num_workers=0)
    model.fit(loader)
    assert np.isfinite(model.epoch_losses[0])


# ------------------------------------------------------------------ BASELINE size (C2) properties
def test_c2_scale_step_properties(ops):
    """U=1M, I=100k, d=64, one 1M-sample step (BASELINE configs[1] shapes): the oracle is too
    slow here, so check size-independent properties against a plain torch fp32 restatement
    of the same closed form on the GPU: total loss, and the updated tables."""
    U, I, d, B = 1_000_000, 100_000, 64, 1 << 20
    gen = torch.Generator(device=DEV)
    gen.manual_seed(2022)
    P = torch.randn(U, d, device=DEV, generator=gen) * 0.01
    Q = torch.randn(I, d, device=DEV, generator=gen) * 0.01
    u = torch.randint(0, U, (B,), device=DEV, generator=gen, dtype=torch.int32)
    i = torch.randint(0, I, (B,), device=DEV, generator=gen, dtype=torch.int32)
    j = torch.randint(0, I, (B,), device=DEV, generator=gen, dtype=torch.int32)
    lr, r1, r2 = 0.01, 1e-3, 1e-3
    ul, il, jl = u.long(), i.long(), j.long()
    pu, qi, qj = P[ul], Q[il], Q[jl]
    x = (pu * qi).sum(-1) - (pu * qj).sum(-1)
    s = torch.sigmoid(x)
    loss = -(1e-10 + s).log().double().sum()
    nU, nI, nJ = (t.double().pow(2).sum().sqrt() for t in (pu, qi, qj))
    loss = loss + r1 * (pu.abs().double().sum() + qi.abs().double().sum() + qj.abs().double().sum()) \
        + r2 * (nU + nI + nJ)
    c = (-(s * (1 - s)) / (1e-10 + s)).unsqueeze(1)
    Pn, Qn = P.clone(), Q.clone()
    Pn.index_add_(0, ul, -lr * (c * (qi - qj) + r1 * pu.sign() + r2 * pu / nU.float()))
    Qn.index_add_(0, il, -lr * (c * pu + r1 * qi.sign() + r2 * qi / nI.float()))
    Qn.index_add_(0, jl, -lr * (-c * pu + r1 * qj.sign() + r2 * qj / nJ.float()))
    for mode in ("fused", "chunked", "atomic", "sorted"):
        P1, Q1 = P.clone(), Q.clone()
        ctx = ops.BprContext(B, d, U, I)
        sl = torch.zeros(1, dtype=torch.float64, device=DEV)
        ctx.set_batch(u, i, j)
        ctx.sgd_step(P1, Q1, lr, r1, r2, item_mode=ops.ITEM_MODES[mode], step_loss=sl)
        torch.cuda.synchronize()
        assert abs(float(sl.cpu()) - float(loss.cpu())) <= 1e-5 * float(loss.cpu())
        assert float((P1 - Pn).abs().max().cpu()) < 1e-6
        assert float((Q1 - Qn).abs().max().cpu()) < 1e-6
        assert float(ctx.gQ.abs().max().cpu()) == 0.0
        # rows not in the batch did not move
        untouched = torch.ones(U, dtype=torch.bool, device=DEV)
        untouched[ul] = False
        assert torch.equal(P1[untouched], P[untouched])
        ctx.close()


def test_sharded_trainer_on_rccl_world1(ops):
    """The multi-GPU step protocols (daisyrec_amd/sharding.py) on the real backend: in a group of one rank
    RCCL's all_reduce / reduce_scatter_tensor / all_gather_into_tensor are identities, so the sharded step
    (staged and phase protocol) must equal the single-GPU step."""
    import os
    import torch.distributed as dist
    from daisyrec_amd.sharding import UserShardedBprTrainer
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29577")
    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        created = True
    try:
        rng = np.random.default_rng(11)
        U, I, d, B = 400, 300, 64, 2048
        P0 = (rng.standard_normal((U, d)) * 0.1).astype(np.float32)
        Q0 = (rng.standard_normal((I, d)) * 0.1).astype(np.float32)
        tri = np.stack([rng.integers(0, U, B), rng.integers(0, I, B), rng.integers(0, I, B)], 1).astype(np.int32)
        loss, Pn, Qn = O.mf_sgd_step(P0, Q0, tri[:, 0], tri[:, 1], tri[:, 2], 0.01, 1e-3, 1e-3)
        for overlap, mode, slices in ((True, "fused", 1), (False, "fused", 1), (True, "chunked", 1), (False, "sorted", 1),
                                      (True, "fused", 4), (True, "fused", 7)):        # 7: 300 items need padding
            P, Q = _t(P0), _t(Q0)
            ctx = ops.BprContext(B, d, U, I)
            tr = UserShardedBprTrainer(ctx, P, Q, 0, 0.01, 1e-3, 1e-3, overlap=overlap,
                                       item_mode=ops.ITEM_MODES[mode], always_collective=True, slices=slices)
            assert tr.collective and tr.staged == (mode == "fused")
            stats = tr.step_from_triples(_t(tri))
            torch.cuda.synchronize()
            assert abs(float(stats[7].cpu()) - loss) <= 1e-5 * abs(loss)
            np.testing.assert_allclose(P.cpu().numpy(), Pn, atol=3e-6)
            np.testing.assert_allclose(Q.cpu().numpy(), Qn, atol=3e-6)
            assert float(ctx.gQ.abs().max().cpu()) == 0.0
            if tr.staged:
                assert float(tr.gQ.abs().max().cpu()) == 0.0 and float(tr.cnt.abs().max().cpu()) == 0.0
            ctx.close()
        # the touched-rows exchange (round 6) over the same backend: 2 B << I, two steps (the second re-uses the re-zeroed
        # gQ / cnt), against the oracle; only rows the batches touched may move
        I2, B2 = 20000, 256
        Q0b = (rng.standard_normal((I2, d)) * 0.1).astype(np.float32)
        tris = [np.stack([rng.integers(0, U, n_), rng.integers(0, I2, n_), rng.integers(0, I2, n_)], 1).astype(np.int32)
                for n_ in (B2, 100)]
        Pn, Qn = P0, Q0b
        P, Q = _t(P0), _t(Q0b)
        ctx = ops.BprContext(B2, d, U, I2)
        tr = UserShardedBprTrainer(ctx, P, Q, 0, 0.01, 1e-3, 1e-3, item_mode=ops.ITEM_MODES["fused"], always_collective=True,
                                   slices=4, exchange="sparse")
        assert tr.sparse and tr.slices == 1 and tr.cap_local == 2 * B2 and tr.wire_bytes["used"] == "sparse"
        for tri2 in tris:
            loss, Pn, Qn = O.mf_sgd_step(Pn, Qn, tri2[:, 0], tri2[:, 1], tri2[:, 2], 0.01, 1e-3, 1e-3)
            stats = tr.step_from_triples(_t(tri2))
            torch.cuda.synchronize()
            assert abs(float(stats[7].cpu()) - loss) <= 1e-5 * abs(loss)
            assert float(tr.gQ.abs().max().cpu()) == 0.0 and float(tr.cnt.abs().max().cpu()) == 0.0
        np.testing.assert_allclose(P.cpu().numpy(), Pn, atol=3e-6)
        np.testing.assert_allclose(Q.cpu().numpy(), Qn, atol=3e-6)
        moved = (Q.cpu().numpy() != Q0b).any(1)
        hit = np.zeros(I2, bool)
        for tri2 in tris:
            hit[tri2[:, 1]] = True
            hit[tri2[:, 2]] = True
        assert np.array_equal(moved, hit)
        ctx.close()
    finally:
        if created:
            dist.destroy_process_group()


@pytest.mark.parametrize("U,B", [(7, 3000), (40, 3000), (300, 3000), (100000, 2500), (3, 64)])
def test_user_chunked_long_and_short_runs(ops, U, B):
    """k_user_chunked / k_user_edges: user runs inside a lane group, across groups, across
    chunks (a user filling several whole chunks when U is tiny) must all reduce to the same
    update as the oracle."""
    rng = np.random.default_rng(U + B)
    I, d = 50, 64
    P0 = (rng.standard_normal((U, d)) * 0.1).astype(np.float32)
    Q0 = (rng.standard_normal((I, d)) * 0.1).astype(np.float32)
    u = rng.integers(0, U, B).astype(np.int32)
    i = rng.integers(0, I, B).astype(np.int32)
    j = rng.integers(0, I, B).astype(np.int32)
    loss, Pn, Qn = O.mf_sgd_step(P0, Q0, u, i, j, 0.003, 1e-3, 1e-3)
    P, Q = _t(P0), _t(Q0)
    ctx = ops.BprContext(B, d, U, I)
    ctx.set_batch(_t(u), _t(i), _t(j))
    ctx.sgd_step(P, Q, 0.003, 1e-3, 1e-3, item_mode=ops.ITEM_MODES["chunked"])
    np.testing.assert_allclose(P.cpu().numpy(), Pn, rtol=0, atol=5e-6)
    np.testing.assert_allclose(Q.cpu().numpy(), Qn, rtol=0, atol=5e-6)
    ctx.close()


def test_build_candidates_bit_exact_and_semantics(ops, ml100k):
    """build_candidates_set (utils.py:53-85) on the device: bit-exact vs the oracle, and the
    reference's semantics (negatives never in test or train rows, truths at the tail, the
    >cand_num-truths branch)."""
    from daisyrec_amd.utils.utils import build_candidates_set
    g = ml100k
    U, I = int(g["user_num"]), int(g["item_num"])
    tr_u, tr_i = g["train_users"], g["train_items"]
    test_u = g["test_u"]
    # a synthetic test split: for every test user the reference's truths are the tail of its candidates
    truths = {}
    rng = np.random.default_rng(0)
    train_sets = {}
    for a, b in zip(tr_u, tr_i):
        train_sets.setdefault(int(a), set()).add(int(b))
    for u in test_u[:60]:
        free = np.setdiff1d(np.arange(I), list(train_sets.get(int(u), ())))
        truths[int(u)] = set(rng.choice(free, size=int(rng.integers(1, 30)), replace=False).tolist())
    big = int(test_u[60])
    free = np.setdiff1d(np.arange(I), list(train_sets.get(big, ())))
    truths[big] = set(free[:40].tolist())                       # more truths than cand_num=25 below
    te_u = np.array([u for u, s in truths.items() for _ in s], dtype=np.int32)
    te_i = np.array([i for _, s in truths.items() for i in s], dtype=np.int32)
    ip_te_o, it_te_o = _csr(te_u, te_i, U)
    ip_tr_o, it_tr_o = _csr(tr_u, tr_i, U)
    ip_te, it_te = ops.build_user_csr(_t(te_u), _t(te_i), U)
    ip_tr, it_tr = ops.build_user_csr(_t(tr_u), _t(tr_i), U)
    users = np.array(list(truths.keys()), dtype=np.int64)
    for cand_num in (25, 100):
        got = ops.build_candidates(ip_te, it_te, ip_tr, it_tr, _t(users), I, cand_num, 7).cpu().numpy()
        want = O.build_candidates(ip_te_o, it_te_o, ip_tr_o, it_tr_o, users, I, cand_num, 7)
        np.testing.assert_array_equal(got, want)
        for row, u in enumerate(users):
            t = sorted(truths[int(u)])
            if len(t) > cand_num:
                assert set(got[row].tolist()) <= set(t)
                continue
            n_neg = cand_num - len(t)
            assert got[row, n_neg:].tolist() == t
            negs = set(got[row, :n_neg].tolist())
            assert not (negs & set(t)) and not (negs & train_sets.get(int(u), set()))
    cfg = mf_config(user_num=U, item_num=I, cand_num=100)
    tu, tc = build_candidates_set(truths, train_sets, cfg)
    assert tu == list(truths.keys()) and len(tc) == len(tu) and tc[3][1].shape == (100,)
    assert tc[0][0] == tu[0] and tc[0][1][-1] == max(truths[tu[0]])


def test_fit_device_shuffle_trains(ml100k):
    """config['shuffle_mode']='device' (Feistel order, throughput kernels): not the reference's batch
    stream, so no golden to match, but the same optimisation: the epoch losses must fall and land
    close to the reference's first epochs."""
    from daisyrec_amd.model.MFRecommender import MF
    from daisyrec_amd.utils.dataset import BasicDataset, get_dataloader
    g = ml100k
    cfg = mf_config(user_num=int(g["user_num"]), item_num=int(g["item_num"]), epochs=3,
                    item_mode="chunked", shuffle_mode="device")
    torch.manual_seed(int(g["seed"]))
    model = MF(cfg)
    loader = get_dataloader(BasicDataset(g["samples"]), batch_size=256, shuffle=True, num_workers=0)
    model.fit(loader)
    L = model.epoch_losses
    assert L[0] > L[1] > L[2]
    for got, ref in zip(L, g["epoch_losses"]):
        assert abs(got - ref) <= 0.02 * ref            # different shuffles of the same data


def test_pointwise_cl_through_sampler_and_fit(ops, ml100k):
    """CL loss end to end through the mirrors (sampler.py:93-98 layout, MFRecommender.py:75-81):
    the device sampler emits (u, i, 1) rows followed by (u, neg, 0) rows; MF.fit trains on them and
    one collated batch gives the oracle's point-wise loss."""
    import pandas as pd
    from daisyrec_amd.model.MFRecommender import MF
    from daisyrec_amd.utils.dataset import BasicDataset, get_dataloader
    from daisyrec_amd.utils.sampler import BasicNegtiveSampler
    g = ml100k
    U, I = int(g["user_num"]), int(g["item_num"])
    df = pd.DataFrame({"user": g["train_users"], "item": g["train_items"], "rating": 1})
    cfg = mf_config(user_num=U, item_num=I, num_ng=2, loss_type="CL", epochs=2, item_mode="chunked")
    rows = BasicNegtiveSampler(df, cfg).sampling()
    n = len(df)
    assert rows.shape == (3 * n, 3) and rows.dtype == np.int32
    np.testing.assert_array_equal(rows[:n], np.stack([g["train_users"], g["train_items"], np.ones(n, np.int32)], 1))
    assert (rows[n:, 2] == 0).all()
    np.testing.assert_array_equal(rows[n::1][::2, 0][:n], g["train_users"])
    pos = {(int(a), int(b)) for a, b in zip(g["train_users"], g["train_items"])}
    assert not any((int(a), int(b)) in pos for a, b in rows[n:n + 5000, :2])
    torch.manual_seed(3)
    model = MF(cfg)
    P0 = model.embed_user.weight.detach().numpy().copy()
    Q0 = model.embed_item.weight.detach().numpy().copy()
    b = rows[np.random.default_rng(0).permutation(len(rows))[:512]]
    want = O.mf_point_grad(P0, Q0, b[:, 0], b[:, 1], b[:, 2], cfg["reg_1"], cfg["reg_2"], O.LOSS_CL)[0]
    got = float(model.calc_loss([torch.from_numpy(b[:, k].copy()) for k in range(3)]).cpu())
    assert abs(got - want) <= 1e-5 * abs(want)
    loader = get_dataloader(BasicDataset(rows), batch_size=1024, shuffle=True, num_workers=0)
    model.fit(loader)
    assert model.epoch_losses[1] < model.epoch_losses[0] and np.isfinite(model.epoch_losses).all()
    with pytest.raises(ValueError):                 # pair-wise loss on a point-wise batch layout
        ctx = ops.BprContext(8, 32, U, I)
        ctx.set_pointwise(True)
        z = _t(np.zeros(8, np.int32))
        ctx.set_batch(z, z, z)
        ctx.forward(model.embed_user.weight.data, model.embed_item.weight.data, ops.LOSS_IDS["BPR"])


def test_chunked_mode_is_bitwise_reproducible(ops):
    """The throughput kernels leave no summation order to chance (parked partials added in group order,
    edge records chained in chunk order, fixed-order batch sums): two runs give identical bits - on a batch
    with hot users and hot items whose runs cross many chunks."""
    rng = np.random.default_rng(3)
    U, I, d, B = 5000, 300, 64, 60000
    P0 = (rng.standard_normal((U, d)) * 0.1).astype(np.float32)
    Q0 = (rng.standard_normal((I, d)) * 0.1).astype(np.float32)
    u = np.where(rng.random(B) < 0.3, 7, rng.integers(0, U, B)).astype(np.int32)       # a user with ~18k samples
    i = np.where(rng.random(B) < 0.3, 5, rng.integers(0, I, B)).astype(np.int32)       # an item with ~18k entries
    j = rng.integers(0, I, B).astype(np.int32)
    outs = []
    for rep in range(3):
        P, Q = _t(P0), _t(Q0)
        ctx = ops.BprContext(B, d, U, I)
        sl = torch.zeros(1, dtype=torch.float64, device=DEV)
        ctx.set_batch(_t(u), _t(i), _t(j))
        for _ in range(3):
            ctx.sgd_step(P, Q, 1e-3, 1e-3, 1e-3, item_mode=ops.ITEM_MODES["chunked"], step_loss=sl)
        outs.append((float(sl.cpu()), P.cpu().numpy(), Q.cpu().numpy()))
        ctx.close()
    for o in outs[1:]:
        assert o[0] == outs[0][0] and np.array_equal(o[1], outs[0][1]) and np.array_equal(o[2], outs[0][2])
    loss, Pn, Qn = P0, Q0, None
    Pn, Qn = P0, Q0
    for _ in range(3):
        loss, Pn, Qn = O.mf_sgd_step(Pn, Qn, u, i, j, 1e-3, 1e-3, 1e-3)
    assert abs(outs[0][0] - loss) <= 1e-5 * abs(loss)
    np.testing.assert_allclose(outs[0][1], Pn, atol=2e-5)
    np.testing.assert_allclose(outs[0][2], Qn, atol=2e-5)


def test_c2_scale_step_against_the_numpy_oracle(ops):
    """BASELINE configs[1] at full size against the ORACLE itself (the test above compares with a torch
    restatement on the GPU, the oracle link at this size was transitive): U = 1 M, I = 100 K, d = 64, one 2 097 152-sample
    step - the bench's headline step: the first batch of a device-shuffled epoch of the partitioned plan through the
    staged kernels - against oracle.mf_sgd_step (numpy, float64, the dense gradients of MFRecommender.py:70-97 + SGD) on
    the same rows.  ~45 s of numpy on the host."""
    U, I, d, B, n = 1_000_000, 100_000, 64, 1 << 21, 5_000_000
    gen = torch.Generator(device=DEV)
    gen.manual_seed(99)
    P = torch.randn(U, d, device=DEV, generator=gen) * 0.01
    Q = torch.randn(I, d, device=DEV, generator=gen) * 0.01
    u = torch.sort(torch.randint(0, U, (n,), device=DEV, generator=gen)).values
    tri = torch.stack([u, torch.randint(0, I, (n,), device=DEV, generator=gen),
                       torch.randint(0, I, (n,), device=DEV, generator=gen)], 1).to(torch.int32).contiguous()
    index = ops.TrainIndex(tri, U, I, user_sorted=True)
    plan = ops.EpochPlan(n, U, I).build_indexed(index, B, order="feistel", seed=5, epoch=1)
    rows = torch.stack(plan.read_batch(0, B)[:3], 1).cpu().numpy().astype(np.int64)
    assert rows.shape == (B, 3)
    P0, Q0 = P.cpu().numpy(), Q.cpu().numpy()
    lr, r1, r2 = 0.01, 1e-3, 1e-3
    ctx = ops.BprContext(B, d, U, I)
    sl = torch.zeros(1, dtype=torch.float64, device=DEV)
    ctx.set_batch_from_plan(plan, 0)
    ctx.sgd_step(P, Q, lr, r1, r2, item_mode=ops.ITEM_MODES["fused"], step_loss=sl)
    torch.cuda.synchronize()
    want, Pn, Qn = O.mf_sgd_step(P0, Q0, rows[:, 0], rows[:, 1], rows[:, 2], lr, r1, r2)
    got = float(sl.cpu())
    assert abs(got - want) <= 1e-5 * abs(want), (got, want)          # (north_star's bar; measured ~1e-9)
    assert abs(got - want) <= 1e-7 * abs(want), (got, want)
    dP = np.abs(P.cpu().numpy() - Pn)
    dQ = np.abs(Q.cpu().numpy() - Qn)
    # fp32 sums of ~2 (users) / ~42 (items) terms of 1e-4 against the float64 result rounded once: a few ulps of 0.01
    print(f"c2-scale step vs oracle: loss rel {abs(got - want) / abs(want):.2e}, max |dP| {dP.max():.2e}, max |dQ| {dQ.max():.2e}")
    assert dP.max() < 3e-8 and dQ.max() < 1e-7, (dP.max(), dQ.max())
    touched = np.zeros(U, bool)
    touched[rows[:, 0]] = True
    assert np.array_equal(P.cpu().numpy()[~touched], P0[~touched])
    ctx.close(); plan.close(); index.close()
