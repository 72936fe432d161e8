"""The oracle against the golden vectors the REAL reference produced
(tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import bpr_mf_numpy as O


def _run_case(g, name, dtype):
    U, I, d, B, ns = g[f"{name}/meta"]
    lr, r1, r2 = g[f"{name}/hyper"]
    lt = O.LOSS_IDS[str(g[f"{name}/loss_type"])]
    opt = str(g[f"{name}/optimizer"])
    P, Q = g[f"{name}/P0"], g[f"{name}/Q0"]
    adam = O.DenseAdam([P.shape, Q.shape], lr, dtype=dtype) if opt == "adam" else None
    for s in range(ns):
        u, i, j = g[f"{name}/u"][s], g[f"{name}/i"][s], g[f"{name}/j"][s]
        if adam is None:
            loss, P, Q = O.mf_sgd_step(P, Q, u, i, j, lr, r1, r2, lt, dtype=dtype)
        else:
            loss, gP, gQ = O.mf_pair_grad(P, Q, u, i, j, r1, r2, lt, dtype=dtype)
            P, Q = adam.step([P, Q], [gP, gQ])
        yield s, loss, P, Q


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_kat_steps(kat_steps, dtype):
    g = kat_steps
    for name in g["names"]:
        name = str(name)
        for s, loss, P, Q in _run_case(g, name, dtype):
            ref_loss = g[f"{name}/loss"][s]
            assert abs(loss - ref_loss) <= 2e-6 * abs(ref_loss), (name, s)
            scale = max(1.0, float(np.abs(g[f"{name}/P0"]).max()))
            np.testing.assert_allclose(P, g[f"{name}/P"][s], rtol=0, atol=1e-6 * scale, err_msg=name)
            np.testing.assert_allclose(Q, g[f"{name}/Q"][s], rtol=0, atol=1e-6 * scale, err_msg=name)


def test_rank_kat(rank_kat):
    r = rank_kat
    pred, _ = O.mf_rank(r["P"], r["Q"], r["us"], r["cands"], int(r["topk"]))
    assert pred.dtype == np.float32
    np.testing.assert_array_equal(pred, r["preds"])
    full = np.stack([O.mf_full_rank(r["P"], r["Q"], int(u), int(r["topk"])) for u in r["us"]])
    np.testing.assert_array_equal(full, r["full"])
    pp = O.mf_forward(r["P"], r["Q"], r["us"], r["cands"][:, 0])
    np.testing.assert_allclose(pp, r["predict"], rtol=1e-5, atol=1e-7)


def ml100k_epoch_orders(g):
    """Batch order of the reference run, from the saved torch RNG state."""
    n = len(g["samples"])
    torch.set_rng_state(torch.from_numpy(g["rng_state_before_fit"]))
    for _ in range(int(g["epochs"])):
        torch.empty((), dtype=torch.int64).random_()                      # _base_seed
        seed = int(torch.empty((), dtype=torch.int64).random_().item())   # RandomSampler seed
        gen = torch.Generator()
        gen.manual_seed(seed)
        yield torch.randperm(n, generator=gen).numpy()


def test_ml100k_end_to_end(ml100k):
    """BASELINE config C1: epoch losses within 1e-5 (relative) and identical top-N."""
    g = ml100k
    samples, B = g["samples"], int(g["batch_size"])
    lr, r1, r2 = g["hyper"]
    P, Q = g["P0"].copy(), g["Q0"].copy()
    for ep, perm in enumerate(ml100k_epoch_orders(g)):
        tot = 0.0
        for s in range(0, len(samples), B):
            idx = perm[s:s + B]
            loss, P, Q = O.mf_sgd_step(P, Q, samples[idx, 0], samples[idx, 1], samples[idx, 2],
                                       lr, r1, r2)
            tot += loss
        ref = g["epoch_losses"][ep]
        assert abs(tot - ref) <= 1e-5 * abs(ref)
    np.testing.assert_allclose(P, g["P1"], atol=2e-4)
    np.testing.assert_allclose(Q, g["Q1"], atol=2e-4)
    pred, _ = O.mf_rank(P, Q, g["test_u"], g["cands"], int(g["topk"]))
    np.testing.assert_array_equal(pred, g["preds"])
    full = np.stack([O.mf_full_rank(P, Q, int(u), int(g["topk"])) for u in g["test_u"][:16]])
    np.testing.assert_array_equal(full, g["full_rank16"])


def test_ml100k_default_run_end_to_end():
    """The run test.py does when nothing is overridden (factors 100, num_ng 4, B=256; mf.yaml, basic.yaml):
    the oracle over the reference's own triples / init / batch order."""
    from conftest import GOLDEN
    import os
    g = np.load(os.path.join(GOLDEN, "ml100k_default.npz"))
    samples, B = g["samples"], int(g["batch_size"])
    lr, r1, r2 = g["hyper"]
    P, Q = g["P0"].copy(), g["Q0"].copy()
    for ep, perm in enumerate(ml100k_epoch_orders(g)):
        tot = 0.0
        for s in range(0, len(samples), B):
            idx = perm[s:s + B]
            loss, P, Q = O.mf_sgd_step(P, Q, samples[idx, 0], samples[idx, 1], samples[idx, 2], lr, r1, r2)
            tot += loss
        ref = g["epoch_losses"][ep]
        assert abs(tot - ref) <= 1e-5 * abs(ref)
    np.testing.assert_allclose(P, g["P1"], atol=5e-4)
    np.testing.assert_allclose(Q, g["Q1"], atol=5e-4)
    pred, _ = O.mf_rank(P, Q, g["test_u"], g["cands"], int(g["topk"]))
    assert (pred == g["preds"]).all(1).mean() >= 0.97


# ---- sampler -----------------------------------------------------------------
def test_philox_known_answers():
    """Random123 kat_vectors for philox4x32-10."""
    assert O.philox4x32_10((0, 0, 0, 0), (0, 0)) == (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)
    f = 0xFFFFFFFF
    assert O.philox4x32_10((f, f, f, f), (f, f)) == (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)
    assert O.philox4x32_10((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344),
                           (0xa4093822, 0x299f31d0)) == (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)


def test_kth_in_complement_exhaustive():
    rng = np.random.default_rng(0)
    for _ in range(50):
        I = int(rng.integers(1, 40))
        row = np.sort(rng.choice(I, size=int(rng.integers(0, I + 1)), replace=False))
        comp = np.setdiff1d(np.arange(I), row)
        for r, want in enumerate(comp):
            assert O.kth_in_complement(row, r) == want


def test_sampler_semantics(ml100k):
    """sampler.py:84-89 semantics on the ml-100k train set: never a train positive,
    all ids (even users without train rows) get negatives, empirical distribution uniform."""
    g = ml100k
    U, I = int(g["user_num"]), int(g["item_num"])
    users, items = g["train_users"].astype(np.int64), g["train_items"]
    order = np.lexsort((items, users))
    indptr = np.zeros(U + 1, dtype=np.int64)
    np.add.at(indptr, users + 1, 1)
    indptr = np.cumsum(indptr)
    csr = items[order]
    js = O.sample_uniform_neg_per_user(indptr, csr, I, 4, seed=2022)
    assert js.shape == (U, 4) and js.min() >= 0 and js.max() < I
    for u in range(U):
        row = set(csr[indptr[u]:indptr[u + 1]].tolist())
        assert not (set(js[u].tolist()) & row)
    tri = O.expand_triples(users, items, js)
    assert tri.shape == (len(users) * 4, 3) and tri.dtype == np.int32
    np.testing.assert_array_equal(tri[::4, 0], users)
    np.testing.assert_array_equal(tri[1::4, 1], items)
    np.testing.assert_array_equal(tri[:, 2].reshape(-1, 4), js[users])
    # uniformity for one user with many draws (chi-square, 5 sigma)
    u = int(np.argmax(np.diff(indptr)))
    row = csr[indptr[u]:indptr[u + 1]]
    ip1 = np.array([0, len(row)])
    draws = O.sample_uniform_neg_per_user(ip1, row, I, 20000, seed=7)[0]
    comp = np.setdiff1d(np.arange(I), row)
    cnt = np.bincount(draws, minlength=I)[comp]
    assert cnt.sum() == 20000
    exp = 20000 / len(comp)
    chi2 = ((cnt - exp) ** 2 / exp).sum()
    dof = len(comp) - 1
    assert abs(chi2 - dof) < 5 * np.sqrt(2 * dof)


def test_torch_port_matches_golden(kat_steps):
    """oracle/torch_port.py (bench.py's cpu_baseline) is the same computation as the reference."""
    from oracle.torch_port import TorchMFBPR
    g, name = kat_steps, "bpr_d64"
    U, I, d, B, ns = (int(x) for x in g[f"{name}/meta"])
    lr, r1, r2 = (float(x) for x in g[f"{name}/hyper"])
    m = TorchMFBPR(U, I, d, lr, r1, r2)
    with torch.no_grad():
        m.embed_user.weight.copy_(torch.from_numpy(g[f"{name}/P0"]))
        m.embed_item.weight.copy_(torch.from_numpy(g[f"{name}/Q0"]))
    for s in range(ns):
        u, i, j = (torch.from_numpy(g[f"{name}/{k}"][s]).long() for k in "uij")
        loss = m.step(u, i, j)
        assert abs(loss - g[f"{name}/loss"][s]) <= 1e-6 * abs(g[f"{name}/loss"][s])
        np.testing.assert_allclose(m.embed_user.weight.detach().numpy(), g[f"{name}/P"][s], atol=1e-7)
        np.testing.assert_allclose(m.embed_item.weight.detach().numpy(), g[f"{name}/Q"][s], atol=1e-7)


def test_adagrad_rmsprop_restatements_match_the_reference():
    """oracle.DenseAdagrad / DenseRMSprop against optim.Adagrad / optim.RMSprop driven by the reference's MF
    (tests/golden/kat_optimizers.npz, make_golden.py::make_kat_optimizers); and what the reference's
    'sparse_adam' branch actually does."""
    import os
    from conftest import GOLDEN
    g = np.load(os.path.join(GOLDEN, "kat_optimizers.npz"))
    assert "SparseAdam does not support dense gradients" in str(g["sparse_adam_error"])
    for name in g["names"]:
        U, I, d, B, ns = (int(x) for x in g[f"{name}/meta"])
        lr, r1, r2 = (float(x) for x in g[f"{name}/hyper"])
        lt = O.LOSS_IDS[str(g[f"{name}/loss_type"])]
        cls = O.DenseAdagrad if str(g[f"{name}/optimizer"]) == "adagrad" else O.DenseRMSprop
        opt = cls([(U, d), (I, d)], lr)
        P, Q = g[f"{name}/P0"], g[f"{name}/Q0"]
        for s in range(ns):
            loss, gP, gQ = O.mf_pair_grad(P, Q, g[f"{name}/u"][s], g[f"{name}/i"][s], g[f"{name}/j"][s], r1, r2, lt)
            assert abs(loss - float(g[f"{name}/loss"][s])) <= 2e-6 * abs(loss)
            P, Q = opt.step([P, Q], [gP, gQ])
            np.testing.assert_allclose(P, g[f"{name}/P"][s], atol=3e-6)
            np.testing.assert_allclose(Q, g[f"{name}/Q"][s], atol=3e-6)


def test_lazy_adam_replay_is_the_dense_sequence():
    """The idea behind ops.LazyAdam, on the oracle's DenseAdam: a row's Adam sequence depends only on its own gradients,
    so leaving rows without a gradient behind and replaying their zero-gradient steps later (per-step constants
    lr/(1-b1^s), sqrt(1-b2^s)) gives exactly the dense optimiser's tables and moments - same operations in the same
    order per row, hence bit-identical in any precision."""
    rng = np.random.default_rng(3)
    U, d, steps, lr, b1, b2, eps = 40, 5, 30, 0.01, 0.9, 0.999, 1e-8
    W0 = rng.standard_normal((U, d)).astype(np.float32)
    touched = [rng.choice(U, size=6, replace=False) for _ in range(steps)]
    grads = [rng.standard_normal((6, d)).astype(np.float32) for _ in range(steps)]

    def one(w, m, v, g, s):                                 # one row, one step (float32 like the kernels)
        f = np.float32
        m = f(b1) * m + f(1 - b1) * g                       # (the exact expression does not matter, only that both
        v = f(b2) * v + f(1 - b2) * g * g                   #  paths use the same one)
        denom = np.sqrt(v) / f(np.sqrt(1.0 - b2 ** s)) + f(eps)
        return (w - f(lr / (1.0 - b1 ** s)) * (m / denom)).astype(f), m.astype(f), v.astype(f)

    # dense: every row every step
    Wd, Md, Vd = W0.copy(), np.zeros_like(W0), np.zeros_like(W0)
    for s in range(1, steps + 1):
        G = np.zeros_like(W0)
        G[touched[s - 1]] = grads[s - 1]
        for r in range(U):
            Wd[r], Md[r], Vd[r] = one(Wd[r], Md[r], Vd[r], G[r], s)
    # lazy: touched rows catch up, take the step; flush at the end
    Wl, Ml, Vl, last = W0.copy(), np.zeros_like(W0), np.zeros_like(W0), np.zeros(U, dtype=np.int64)
    zero = np.zeros(d, dtype=np.float32)
    for s in range(1, steps + 1):
        for k, r in enumerate(touched[s - 1]):
            for q in range(last[r] + 1, s):
                Wl[r], Ml[r], Vl[r] = one(Wl[r], Ml[r], Vl[r], zero, q)
            Wl[r], Ml[r], Vl[r] = one(Wl[r], Ml[r], Vl[r], grads[s - 1][k], s)
            last[r] = s
    for r in range(U):
        for q in range(last[r] + 1, steps + 1):
            Wl[r], Ml[r], Vl[r] = one(Wl[r], Ml[r], Vl[r], zero, q)
    assert np.array_equal(Wd, Wl) and np.array_equal(Md, Ml) and np.array_equal(Vd, Vl)
    # and the dense restatement the goldens pin agrees with this row-wise form to round-off
    opt = O.DenseAdam([W0.shape], lr)
    W = W0.copy()
    for s in range(1, steps + 1):
        G = np.zeros_like(W0)
        G[touched[s - 1]] = grads[s - 1]
        (W,) = opt.step([W], [G])
    np.testing.assert_allclose(W, Wd, rtol=0, atol=2e-6)


def test_partitioned_plan_restatement_from_first_principles():
    """The partitioned epoch plan is this repo's construction (the reference has a DataLoader): its oracle restatement
    is checked against what dataset.py:5-27 defines - batch k serves exactly the rows at epoch positions [kB, (k+1)B) -
    and against the layout the kernels rely on: samples of a batch grouped by user in a stable order, its entries sorted
    by (item, slot) in a stable order, every entry pointing at a sample of its own batch.  Pairwise and point-wise rows."""
    import numpy as np
    from oracle import bpr_mf_numpy as O
    rng = np.random.default_rng(17)
    for n, B, U, I, pointwise in ((1, 1, 1, 1, False), (57, 8, 9, 7, False), (300, 64, 20, 31, True), (1000, 333, 50, 40, False)):
        tri = np.stack([rng.integers(0, U, n), rng.integers(0, I, n),
                        rng.integers(0, 2, n) if pointwise else rng.integers(0, I, n)], 1)
        pos = rng.permutation(n)
        samples, spos, ekey, epos = O.partitioned_plan(tri, pos, B, pointwise=pointwise)
        epl = 1 if pointwise else 2
        assert len(samples) == n and len(ekey) == epl * n
        nb = (n + B - 1) // B
        for k in range(nb):
            lo, hi = k * B, min((k + 1) * B, n)
            rows, rp = samples[lo:hi], spos[lo:hi]
            assert np.array_equal(np.sort(rp), np.arange(lo, hi))                 # exactly the loader's batch k
            by_pos = {int(p): tuple(int(x) for x in tri[t]) for t, p in enumerate(pos)}
            assert all(tuple(int(x) for x in r) == by_pos[int(p)] for r, p in zip(rows, rp))
            assert np.all(np.diff(rows[:, 0]) >= 0)                               # grouped by user ...
            csr = np.argsort(tri[:, 0], kind="stable")                            # ... ties in CSR (array) order
            rank_in_csr = np.empty(n, np.int64)
            rank_in_csr[csr] = np.arange(n)
            t_of = {int(p): t for t, p in enumerate(pos)}
            r_csr = np.array([rank_in_csr[t_of[int(p)]] for p in rp])
            assert np.all(np.diff(r_csr) > 0)
            ek, ep = ekey[epl * lo:epl * hi], epos[epl * lo:epl * hi]
            assert np.all(np.diff(ek) >= 0)                                       # sorted by item << 1 | slot
            assert np.all((ep >= lo) & (ep < hi))                                 # entries point into their own batch
            want = []                                                             # the multiset of (key, position)
            for r, p in zip(rows, rp):
                want.append((int(r[1]) << 1, int(p)))
                if not pointwise:
                    want.append(((int(r[2]) << 1) | 1, int(p)))
            assert sorted(want) == sorted(zip(ek.tolist(), ep.tolist()))


def test_feistel_positions_is_a_keyed_bijection_for_every_size_class():
    """the device shuffle's restatement: a permutation of 0..n-1 for even and odd bit counts of the network's domain,
    different per (seed, epoch), the same for the same key"""
    import numpy as np
    from oracle import bpr_mf_numpy as O
    for n in (1, 2, 3, 4, 5, 8, 9, 31, 33, 511, 513, 4096, 4097, 100003):
        p = O.feistel_positions(n, 5, 1)
        assert np.array_equal(np.sort(p), np.arange(n)), n
        assert np.array_equal(p, O.feistel_positions(n, 5, 1))
        if n > 8:
            assert not np.array_equal(p, O.feistel_positions(n, 5, 2)) and not np.array_equal(p, O.feistel_positions(n, 6, 1))
