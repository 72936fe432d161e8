"""Randomised tests of the two widened models' kernels against their oracles: LightGCN's normalised adjacency product,
propagation and its transpose (csrc/lightgcn.hip; LightGCNRecommender.py:74-129) on random bipartite graphs - isolated
nodes, hubs with thousands of neighbours, duplicate interactions, any d and depth -, and FM's biases on the staged epoch
(FMRecommender.py:61-68) at random sizes, losses and skews.  Case k is a pure function of (DAISY_FUZZ_SEED, k);
DAISY_FUZZ_CASES widens the campaign (profiles/r05_fuzz.txt)."""
import os

import numpy as np
import pytest
import torch

from oracle import fm_numpy as F
from oracle import lightgcn_numpy as LG

pytestmark = pytest.mark.gpu
DEV = "cuda"
CASES = int(os.environ.get("DAISY_FUZZ_CASES", "24"))
SEED = int(os.environ.get("DAISY_FUZZ_SEED", "2022"))


def _t(x):
    return torch.as_tensor(np.ascontiguousarray(x)).to(DEV)


def _log_uniform(rng, lo, hi):
    return int(round(float(np.exp(rng.uniform(np.log(lo), np.log(hi))))))


def _ids(rng, n, size, alpha):
    if alpha == 0.0 or size == 1:
        return rng.integers(0, size, n)
    w = 1.0 / np.arange(1, size + 1) ** alpha
    return rng.permutation(size)[rng.choice(size, n, p=w / w.sum())]


@pytest.mark.parametrize("k", range(CASES))
def test_random_graph_products_match_the_oracle(k):
    from daisyrec_amd import ops
    rng = np.random.default_rng([SEED, k, 4])
    U, I = _log_uniform(rng, 1, 8000), _log_uniform(rng, 1, 8000)
    n = _log_uniform(rng, 1, 150_000)
    d = int(rng.choice([4, 8, 20, 32, 50, 64, 100, 128]))
    L = int(rng.integers(1, 5))
    a_u, a_i = float(rng.choice([0.0, 0.0, 1.0, 1.4])), float(rng.choice([0.0, 0.0, 1.0, 1.4]))
    tag = dict(k=k, U=U, I=I, n=n, d=d, L=L, a_u=a_u, a_i=a_i)
    gu, gi = _ids(rng, n, U, a_u), _ids(rng, n, I, a_i)                # (duplicate interactions collapse: :88-90)
    graph = ops.LgcnGraph(_t(gu), _t(gi), U, I)
    ref = LG.norm_adj_csr(gu, gi, U, I)
    X = rng.standard_normal((U + I, d)).astype(np.float32)
    G = rng.standard_normal((U + I, d)).astype(np.float32)
    want = LG.spmm(ref, X.astype(np.float64))
    # a row is an fp32 sum of `deg` products (values <= 1, |x| ~ 1): round-off relative to the sum of their sizes
    deg = np.diff(ref[0]).astype(np.float64)
    scale = LG.spmm((ref[0], ref[1], np.abs(ref[2])), np.abs(X).astype(np.float64)).max(1)
    tol = (2e-6 + 3e-7 * np.sqrt(np.maximum(deg, 1.0)) * scale)[:, None]
    got = graph.spmm(_t(X)).cpu().numpy()
    assert (np.abs(got - want) <= tol).all(), (tag, "spmm", float((np.abs(got - want) / tol).max()))
    want_p = LG.propagate(ref, X.astype(np.float64), L)
    got_p = graph.propagate(_t(X), L).cpu().numpy()
    assert (np.abs(got_p - want_p) <= 2.0 * tol + 2e-6 * np.abs(X)).all(), (tag, "propagate", float(np.abs(got_p - want_p).max()))
    # the transpose (A_hat is symmetric: the same operator), accumulated ON TOP of what dE0 holds
    dE0 = _t(X.copy())
    graph.backprop(_t(G), L, dE0)
    want_b = X.astype(np.float64) + LG.propagate(ref, G.astype(np.float64), L)
    scale_g = LG.spmm((ref[0], ref[1], np.abs(ref[2])), np.abs(G).astype(np.float64)).max(1)
    tol_g = (4e-6 + 6e-7 * np.sqrt(np.maximum(deg, 1.0)) * scale_g)[:, None] + 2e-6 * (np.abs(G) + np.abs(X))
    assert (np.abs(dE0.cpu().numpy() - want_b) <= tol_g).all(), (tag, "backprop")
    # <propagate(X), G> = <X, propagate^T(G)>
    lhs = float((got_p.astype(np.float64) * G).sum())
    rhs = float((X.astype(np.float64) * (dE0.cpu().numpy().astype(np.float64) - X)).sum())
    assert abs(lhs - rhs) <= 1e-5 * (np.abs(got_p.astype(np.float64) * G).sum() + 1.0), (tag, lhs, rhs)
    graph.close()


@pytest.mark.parametrize("k", range(CASES))
def test_random_fm_epoch_matches_the_oracle(k, monkeypatch):
    from daisyrec_amd import ops
    rng = np.random.default_rng([SEED, k, 5])
    d = int(rng.choice([4, 8, 16, 20, 32, 50, 64, 100, 128]))
    B = max(1, min(_log_uniform(rng, 1, 100_000), 4_000_000 // d))
    U, I = _log_uniform(rng, 1, 40_000), _log_uniform(rng, 2, 20_000)
    n = max(1, min(int(B * rng.uniform(1.0, 3.3)), 250_000))
    loss = str(rng.choice(["BPR", "BPR", "HL", "TL", "CL", "SL"]))
    a_u, a_i = float(rng.choice([0.0, 0.0, 0.7, 1.0, 1.3])), float(rng.choice([0.0, 0.0, 0.7, 1.0, 1.3]))
    reg_1, reg_2 = [(0.0, 0.0), (1e-3, 2e-3), (0.01, 0.0), (0.0, 5e-3)][int(rng.integers(0, 4))]
    env = {"DAISY_STAGED_SPARSE": rng.choice([None, None, "0", "1"]), "DAISY_EDGE_BLOCKS": rng.choice([None, None, "0", "1"])}
    tag = dict(k=k, d=d, B=B, U=U, I=I, n=n, loss=loss, a_u=a_u, a_i=a_i, reg=(reg_1, reg_2), env={a: None if b is None else str(b) for a, b in env.items()})
    for key, v in env.items():
        if v is not None:
            monkeypatch.setenv(key, str(v))
    point = loss in ("CL", "SL")
    pos = _ids(rng, n, I, a_i)
    third = rng.integers(0, 2, n) if point else (pos + rng.integers(1, I, n)) % I
    tri = np.stack([_ids(rng, n, U, a_u), pos, third], 1).astype(np.int32)
    w0 = [(rng.standard_normal((U, d)) * 0.1).astype(np.float32), (rng.standard_normal((I, d)) * 0.1).astype(np.float32),
          (rng.standard_normal(U) * 0.1).astype(np.float32), (rng.standard_normal(I) * 0.1).astype(np.float32),
          np.array([0.05], np.float32)]
    hot = max(int(np.bincount(tri[:, 0]).max()), int(np.bincount(tri[:, 1]).max()))
    lr = 0.05 / max(1.0, hot / 50.0) / (max(1.0, d / 32.0) if loss == "SL" else 1.0)
    # bias_ sees EVERY sample of a batch (FMRecommender.py:66): keep its step of the order of its size as well
    lr = min(lr, 0.05 / max(1.0, B / 50.0) * 4.0)
    lid = ops.LOSS_IDS[loss]
    index, plan = ops.TrainIndex(_t(tri), U, I, pointwise=point), ops.EpochPlan(n, U, I)
    plan.build_indexed(index, B, order="feistel", seed=SEED % 1000, epoch=k)
    nb = plan.num_batches
    w = [_t(x.copy()) for x in w0]
    ctx = ops.BprContext(B, d, U, I)
    ctx.set_bias(w[2], w[3], w[4], g_i_bias=torch.zeros(I, device=DEV))
    sl = torch.zeros(nb, dtype=torch.float64, device=DEV)
    ctx.fit_epoch_sgd(plan, w[0], w[1], lr, reg_1, reg_2, loss_type=lid, item_mode=ops.ITEM_MODES["fused"], step_losses=sl)
    torch.cuda.synchronize()
    assert float(ctx.epoch_acc[1].cpu()) == 0.0, tag
    cur = list(w0)
    cnt_u, cnt_i, step = np.zeros(U), np.zeros(I), [0.0] * 5
    for b in range(nb):
        u, i, j = (t.cpu().numpy().astype(np.int64) for t in plan.read_batch(b, B)[:3])
        before = [np.asarray(x, np.float64).copy() for x in cur]
        want, *cur = F.fm_sgd_step(*cur, u, i, j, lr, reg_1, reg_2, loss_type=lid)
        assert abs(float(sl[b].cpu()) - want) <= 2e-5 * abs(want) + 1e-6, (tag, b, float(sl[b].cpu()), want)
        cnt_u = np.maximum(cnt_u, np.bincount(u, minlength=U))
        ci = np.bincount(i, minlength=I)
        cnt_i = np.maximum(cnt_i, ci if point else ci + np.bincount(j, minlength=I))
        for t in range(5):
            a = np.abs(np.asarray(cur[t], np.float64).reshape(before[t].shape) - before[t])
            step[t] = np.maximum(step[t], a.max(1) if a.ndim == 2 else a)
    nB = float(min(B, n))
    for t, (name, cnt) in enumerate((("P", cnt_u), ("Q", cnt_i), ("u_bias", cnt_u), ("i_bias", cnt_i), ("bias_", np.array([nB])))):
        got = w[t].cpu().numpy().astype(np.float64).reshape(np.asarray(cur[t]).shape)
        diff = np.abs(got - np.asarray(cur[t], np.float64))
        # the round-off model of tests/test_gpu_fuzz.py: lr times an fp32 sum of cnt terms against fp64
        tol = 3e-7 + lr * 0.3 * (5e-8 * cnt ** 1.5 + 4e-7 * cnt) + 8e-7 * np.sqrt(cnt) * step[t]
        tol = tol[:, None] if diff.ndim == 2 else tol.reshape(diff.shape)
        assert (diff <= tol).all(), (tag, name, float((diff / tol).max()), float(diff.max()))
    ctx.close(); plan.close(); index.close()
