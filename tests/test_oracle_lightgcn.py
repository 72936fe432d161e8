"""oracle/lightgcn_numpy.py against the golden vectors the REAL reference LightGCN produced
(tests/golden/make_golden_lightgcn.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import bpr_mf_numpy as O
from oracle import lightgcn_numpy as LG
from test_oracle_neumf import assert_params_close

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def kat_lg():
    return np.load(os.path.join(HERE, "golden", "kat_lightgcn.npz"))


def test_norm_adjacency_is_the_reference_matrix(kat_lg):
    g = kat_lg
    U, I, d, L = (int(x) for x in g["rank/meta"])
    indptr, col, val = LG.norm_adj_csr(g["rank/gu"], g["rank/gi"], U, I)
    rows = np.repeat(np.arange(U + I), np.diff(indptr))
    order = np.lexsort((g["rank/adj_col"], g["rank/adj_row"]))
    np.testing.assert_array_equal(rows, g["rank/adj_row"][order])
    np.testing.assert_array_equal(col, g["rank/adj_col"][order])
    np.testing.assert_array_equal(val, g["rank/adj_val"][order])          # bit exact (float64 product -> float32)


def oracle_steps(g, name, dtype=np.float64):
    U, I, d, L, B, ns = (int(x) for x in g[f"{name}/meta"])
    lr, r1, r2 = (float(x) for x in g[f"{name}/hyper"])
    lt = O.LOSS_IDS[str(g[f"{name}/loss_type"])]
    graph = LG.norm_adj_csr(g[f"{name}/gu"], g[f"{name}/gi"], U, I)
    P, Q = g[f"{name}/P0"], g[f"{name}/Q0"]
    adam = O.DenseAdam([P.shape, Q.shape], lr, dtype=dtype) if str(g[f"{name}/optimizer"]) == "adam" else None
    for s in range(ns):
        loss, gP, gQ = LG.lightgcn_grad(graph, P, Q, g[f"{name}/u"][s], g[f"{name}/i"][s], g[f"{name}/j"][s], r1, r2, L,
                                        lt, dtype=dtype)
        if adam is None:
            P = (np.asarray(P, dtype) - lr * gP).astype(np.float32)
            Q = (np.asarray(Q, dtype) - lr * gQ).astype(np.float32)
        else:
            P, Q = adam.step([P, Q], [gP, gQ])
        yield s, loss, P, Q


def test_lightgcn_kat_steps(kat_lg):
    g = kat_lg
    for name in g["names"]:
        name = str(name)
        for s, loss, P, Q in oracle_steps(g, name):
            ref = g[f"{name}/loss"][s]
            assert abs(loss - ref) <= 5e-6 * abs(ref), (name, s, loss, ref)
        is_adam = str(g[f"{name}/optimizer"]) == "adam"
        assert_params_close({"P": P, "Q": Q}, {"P": g[f"{name}/P"], "Q": g[f"{name}/Q"]}, ("P", "Q"), name, 3e-6,
                            adam_lr=float(g[f"{name}/hyper"][0]) if is_adam else None, steps=int(g[f"{name}/meta"][5]),
                            frac=0.99)


def test_lightgcn_rank_kat(kat_lg):
    g = kat_lg
    U, I, d, L = (int(x) for x in g["rank/meta"])
    graph = LG.norm_adj_csr(g["rank/gu"], g["rank/gi"], U, I)
    pred, _ = LG.lightgcn_rank(graph, g["rank/P"], g["rank/Q"], g["rank/us"], g["rank/cands"], int(g["rank/topk"]), L)
    assert (pred == g["rank/preds"]).mean() > 0.97
    full = np.stack([LG.lightgcn_full_rank(graph, g["rank/P"], g["rank/Q"], int(u), int(g["rank/topk"]), L)
                     for u in g["rank/us"]])
    assert (full == g["rank/full"]).mean() > 0.97
    out = LG.propagate(graph, np.concatenate([g["rank/P"], g["rank/Q"]], 0), L)
    pp = np.einsum("bk,bk->b", out[g["rank/us"]], out[U + g["rank/cands"][:, 0]])
    np.testing.assert_allclose(pp, g["rank/predict"], rtol=1e-5, atol=1e-6)


def test_lightgcn_ml100k_end_to_end(kat_lg):
    """run_examples/test.py --algo_name lightgcn on ml-100k (first 50 batches): epoch loss within 1e-5."""
    g = kat_lg
    U, I, d, L = (int(x) for x in g["ml/meta"])
    lr, r1, r2 = (float(x) for x in g["ml/hyper"])
    samples, B = g["ml/samples"], int(g["ml/batch_size"])
    graph = LG.norm_adj_csr(g["ml/train_users"], g["ml/train_items"], U, I)
    P, Q = g["ml/P0"], g["ml/Q0"]
    adam = O.DenseAdam([P.shape, Q.shape], lr)
    n = len(samples)
    torch.set_rng_state(torch.from_numpy(g["ml/rng_state_before_fit"]))
    torch.empty((), dtype=torch.int64).random_()
    gen = torch.Generator()
    gen.manual_seed(int(torch.empty((), dtype=torch.int64).random_().item()))
    perm = torch.randperm(n, generator=gen).numpy()
    tot = 0.0
    for s in range(0, n, B):
        idx = perm[s:s + B]
        loss, gP, gQ = LG.lightgcn_grad(graph, P, Q, samples[idx, 0], samples[idx, 1], samples[idx, 2], r1, r2, L)
        P, Q = adam.step([P, Q], [gP, gQ])
        tot += loss
    ref = g["ml/epoch_losses"][0]
    assert abs(tot - ref) <= 1e-5 * abs(ref), (tot, ref)
    for got, key in ((P, "P1"), (Q, "Q1")):
        err = np.linalg.norm(got.astype(np.float64) - g[f"ml/{key}"]) / np.linalg.norm(g[f"ml/{key}"])
        assert err < 1e-3, (key, err)
    pred, _ = LG.lightgcn_rank(graph, P, Q, g["ml/test_u"], g["ml/cands"], int(g["ml/topk"]), L)
    same = (pred == g["ml/preds"]).all(axis=1).mean()
    assert same > 0.9, f"top-N lists identical for {same:.3f} of the users"
