"""Randomised BIT-EXACT tests of the integer work either side of the step: the partitioned epoch plan (any sizes, both row
kinds, the three orders, several tiles per partition), the uniform negative sampler per user and per interaction
(sampler.py:82-101), the candidate-set builder (utils.py:53-85), and the ranking's top-N (MFRecommender.py:106-123, ids
compared through their scores: exact ties keep the candidate order) - each against the oracle's restatement on the same
seeded inputs.  Case k is a pure function of (DAISY_FUZZ_SEED, k); DAISY_FUZZ_CASES widens the campaign
(profiles/r05_fuzz.txt)."""
import os

import numpy as np
import pytest
import torch

from oracle import bpr_mf_numpy as O

pytestmark = pytest.mark.gpu
DEV = "cuda"
CASES = int(os.environ.get("DAISY_FUZZ_CASES", "24"))
SEED = int(os.environ.get("DAISY_FUZZ_SEED", "2022"))


def _t(x, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(x))
    return (t if dtype is None else t.to(dtype)).to(DEV)


def _log_uniform(rng, lo, hi):
    return int(round(float(np.exp(rng.uniform(np.log(lo), np.log(hi))))))


def _csr(users, items, U):
    order = np.lexsort((items, users))
    indptr = np.zeros(U + 1, dtype=np.int64)
    np.add.at(indptr, users.astype(np.int64) + 1, 1)
    return np.cumsum(indptr), items[order]


@pytest.mark.parametrize("k", range(CASES))
def test_random_plan_is_bit_exact(k, monkeypatch):
    from daisyrec_amd import ops
    rng = np.random.default_rng([SEED, k, 1])
    n = _log_uniform(rng, 1, 300_000)
    B = max(1, min(_log_uniform(rng, 1, 400_000), 2 * n))
    U, I = _log_uniform(rng, 1, 100_000), _log_uniform(rng, 1, 50_000)
    point, sort = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    tiles = int(rng.choice([0, 0, 1, 2, 3, 5]))
    order = str(rng.choice(["identity", "perm", "feistel"]))
    seed, epoch = int(rng.integers(0, 1 << 30)), int(rng.integers(0, 100))
    tag = dict(k=k, n=n, B=B, U=U, I=I, point=point, sort=sort, tiles=tiles, order=order)
    if tiles:
        monkeypatch.setenv("DAISY_PART_TILES", str(tiles))
    tri = np.stack([rng.integers(0, U, n), rng.integers(0, I, n),
                    rng.integers(0, 2, n) if point else rng.integers(0, I, n)], 1).astype(np.int32)
    if sort:
        tri = tri[np.argsort(tri[:, 0], kind="stable")]
    index = ops.TrainIndex(_t(tri), U, I, pointwise=point)
    plan = ops.EpochPlan(n, U, I)
    perm = None
    if order == "identity":
        pos = np.arange(n)
    elif order == "perm":
        perm = torch.randperm(n, generator=torch.Generator().manual_seed(seed))
        pos = np.empty(n, dtype=np.int64)
        pos[perm.numpy()] = np.arange(n)
        perm = perm.to(DEV)
    else:
        pos = O.feistel_positions(n, seed, epoch)
    plan.build_indexed(index, B, order=order, perm=perm, seed=seed, epoch=epoch)
    nb = (n + B - 1) // B
    assert plan.num_batches == nb, tag
    samples, spos, ekey, epos = O.partitioned_plan(tri, pos, B, pointwise=point)
    per = 1 if point else 2
    for b in sorted({0, nb - 1, int(rng.integers(0, nb)), int(rng.integers(0, nb))}):
        lo, hi = b * B, min((b + 1) * B, n)
        u, i, j, ei, es, _ = (t.cpu().numpy() for t in plan.read_batch(b, B))
        assert np.array_equal(np.stack([u, i, j], 1), samples[lo:hi]), (tag, b)
        m = per * (hi - lo)
        assert np.array_equal(ei[:m], ekey[per * lo:per * hi] >> 1), (tag, b)
        want_s = (epos[per * lo:per * hi] - lo) if point else ((epos[2 * lo:2 * hi] - lo) | ((ekey[2 * lo:2 * hi] & 1) << 31))
        assert np.array_equal(es[:m].astype(np.uint32), want_s.astype(np.uint32)), (tag, b)
    plan.close(); index.close()


@pytest.mark.parametrize("k", range(CASES))
def test_random_sampler_and_candidates_are_bit_exact(k):
    from daisyrec_amd import ops
    rng = np.random.default_rng([SEED, k, 2])
    U, I = _log_uniform(rng, 1, 2500), _log_uniform(rng, 2, 3000)
    deg = np.minimum(rng.geometric(1.0 / _log_uniform(rng, 1, max(2, I // 2)), U), I)
    deg[rng.random(U) < 0.05] = 0                                   # users without a row
    if I <= 64:
        deg[rng.random(U) < 0.05] = I                               # users who have seen everything: no negative exists
    users = np.repeat(np.arange(U), deg).astype(np.int32)
    items = np.concatenate([rng.choice(I, m, replace=False) for m in deg] + [np.empty(0, np.int64)]).astype(np.int32)
    tag = dict(k=k, U=U, I=I, nnz=len(users))
    if len(users) == 0:
        return
    shuffle = rng.permutation(len(users))
    users, items = users[shuffle], items[shuffle]
    indptr_o, csr_o = _csr(users, items, U)
    indptr, csr = ops.build_user_csr(_t(users), _t(items), U)
    assert np.array_equal(indptr.cpu().numpy(), indptr_o) and np.array_equal(csr.cpu().numpy(), csr_o), tag
    num_ng, seed, epoch = int(rng.integers(1, 7)), int(rng.integers(0, 1 << 30)), int(rng.integers(0, 50))
    js = ops.sample_neg_per_user(indptr, csr, I, num_ng, seed, epoch).cpu().numpy()
    want = O.sample_uniform_neg_per_user(indptr_o, csr_o, I, num_ng, seed, epoch)
    assert np.array_equal(js, want), tag
    assert ((js == -1).all(1) == (np.diff(indptr_o) == I)).all(), tag
    # per interaction (a fresh negative for every row; rows of users without a complement are left out)
    ok = np.diff(indptr_o)[users] < I
    rows = np.flatnonzero(ok)[:4000]
    if len(rows):
        tri = O.expand_triples(users[rows], items[rows], np.zeros((U, 1), np.int32))
        tri_d = _t(tri)
        ops.resample_neg_per_interaction(indptr, csr, I, tri_d, seed + 1, epoch)
        want = O.sample_uniform_neg_per_interaction(indptr_o, csr_o, users[rows], I, 1, seed + 1, epoch)
        assert np.array_equal(tri_d.cpu().numpy()[:, 2], want[:, 0]), tag
    # candidates: a test split disjoint from the rows above
    t_users = np.flatnonzero((np.diff(indptr_o) < I - 1) & (rng.random(U) < 0.3))[:60]
    if len(t_users) == 0:
        return
    te_u, te_i = [], []
    for u in t_users:
        free = np.setdiff1d(np.arange(I), csr_o[indptr_o[u]:indptr_o[u + 1]])
        pick = rng.choice(free, size=int(rng.integers(1, min(len(free), 40) + 1)), replace=False)
        te_u += [u] * len(pick)
        te_i += pick.tolist()
    te_u, te_i = np.array(te_u, np.int32), np.array(te_i, np.int32)
    ip_te_o, it_te_o = _csr(te_u, te_i, U)
    ip_te, it_te = ops.build_user_csr(_t(te_u), _t(te_i), U)
    cand_num = int(rng.choice([5, 25, 100, 300]))
    got = ops.build_candidates(ip_te, it_te, indptr, csr, _t(t_users.astype(np.int64)), I, cand_num, seed).cpu().numpy()
    want = O.build_candidates(ip_te_o, it_te_o, indptr_o, csr_o, t_users.astype(np.int64), I, cand_num, seed)
    assert np.array_equal(got, want), (tag, cand_num)


@pytest.mark.parametrize("k", range(CASES))
def test_random_rank_matches_the_oracle(k):
    from daisyrec_amd import ops
    rng = np.random.default_rng([SEED, k, 3])
    d = int(rng.choice([4, 8, 20, 32, 50, 64, 100, 128, 200]))
    U, I = _log_uniform(rng, 1, 5000), _log_uniform(rng, 1, 20000)
    nu, C = _log_uniform(rng, 1, 300), _log_uniform(rng, 1, 1500)
    topk = int(rng.integers(1, min(C, 100) + 1))
    tag = dict(k=k, d=d, U=U, I=I, nu=nu, C=C, topk=topk)
    P = (rng.standard_normal((U, d)) * 0.5).astype(np.float32)
    Q = (rng.standard_normal((I, d)) * 0.5).astype(np.float32)
    us = rng.integers(0, U, nu).astype(np.int64)
    cands = rng.integers(0, I, (nu, C)).astype(np.int64)               # (duplicates wherever C is not << I: exact ties)
    want, _ = O.mf_rank(P, Q, us, cands, topk)
    got = ops.mf_rank_topk(_t(P), _t(Q), _t(us), _t(cands), topk).cpu().numpy()
    assert got.shape == (nu, topk), tag
    if np.array_equal(got.astype(np.float32), want):
        return
    # not identical: only where two DIFFERENT candidates score within fp32 round-off of each other (the oracle's dot
    # products are numpy's, the device's its own summation order) - the scores along both lists must agree
    s64 = np.einsum("ud,ucd->uc", P[us].astype(np.float64), Q[cands].astype(np.float64))
    score_of = lambda ids: np.stack([[s64[r][np.flatnonzero(cands[r] == int(x))[0]] for x in ids[r]] for r in range(nu)])
    a, b = score_of(got), score_of(want.astype(np.int64))
    assert np.abs(a - b).max() <= 2e-6 * max(1.0, np.abs(s64).max()), (tag, float(np.abs(a - b).max()))
    assert (got.astype(np.float32) != want).mean() < 0.02, tag
