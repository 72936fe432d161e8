"""oracle/item2vec_numpy.py against the golden vectors the REAL reference Item2Vec produced
(tests/golden/make_golden_item2vec.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import bpr_mf_numpy as O
from oracle import item2vec_numpy as IV
from test_oracle_neumf import assert_params_close

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def kat_i2v():
    return np.load(os.path.join(HERE, "golden", "kat_item2vec.npz"))


def test_item2vec_kat_steps(kat_i2v):
    g = kat_i2v
    for name in g["names"]:
        name = str(name)
        U, I, d, B, ns = (int(x) for x in g[f"{name}/meta"])
        lr = float(g[f"{name}/lr"])
        S = g[f"{name}/S0"]
        is_adam = str(g[f"{name}/optimizer"]) == "adam"
        adam = O.DenseAdam([S.shape], lr) if is_adam else None
        for s in range(ns):
            loss, gS = IV.item2vec_grad(S, g[f"{name}/t"][s], g[f"{name}/c"][s], g[f"{name}/y"][s])
            ref = g[f"{name}/loss"][s]
            assert abs(loss - ref) <= 3e-6 * abs(ref), (name, s, loss, ref)
            S = adam.step([S], [gS])[0] if is_adam else (S.astype(np.float64) - lr * gS).astype(np.float32)
        assert_params_close({"S": S}, {"S": g[f"{name}/S"]}, ("S",), name, 3e-6, adam_lr=lr if is_adam else None,
                            steps=ns, frac=0.99)


def test_item2vec_ml100k_end_to_end(kat_i2v):
    g = kat_i2v
    U, I, d = (int(x) for x in g["ml/meta"])
    lr, B = float(g["ml/lr"]), int(g["ml/batch_size"])
    samples = g["ml/samples"]
    n = len(samples)
    S = g["ml/S0"]
    adam = O.DenseAdam([S.shape], lr)
    torch.set_rng_state(torch.from_numpy(g["ml/rng_state_before_fit"]))
    torch.empty((), dtype=torch.int64).random_()
    gen = torch.Generator()
    gen.manual_seed(int(torch.empty((), dtype=torch.int64).random_().item()))
    perm = torch.randperm(n, generator=gen).numpy()
    tot = 0.0
    for s in range(0, n, B):
        idx = perm[s:s + B]
        loss, gS = IV.item2vec_grad(S, samples[idx, 0], samples[idx, 1], samples[idx, 2])
        S = adam.step([S], [gS])[0]
        tot += loss
    ref = g["ml/epoch_losses"][0]
    assert abs(tot - ref) <= 1e-5 * abs(ref), (tot, ref)
    np.testing.assert_allclose(S, g["ml/S1"], atol=2e-5)
    # user embedding build (:56-59) and the rank path on it
    ur = {}
    for u, i in zip(g["ml/ur_users"], g["ml/ur_items"]):
        ur.setdefault(int(u), set()).add(int(i))
    Uemb = g["ml/Uemb0"].copy()
    for u, row in IV.build_user_embedding(g["ml/S1"], ur, U).items():
        Uemb[u] = row
    np.testing.assert_allclose(Uemb, g["ml/Uemb1"], rtol=1e-5, atol=1e-6)
    pred, _ = O.mf_rank(g["ml/Uemb1"], g["ml/S1"], g["ml/test_u"], g["ml/cands"], int(g["ml/topk"]))
    assert (pred == g["ml/preds"]).all(axis=1).mean() > 0.98
    full = np.stack([O.mf_full_rank(g["ml/Uemb1"], g["ml/S1"], int(u), int(g["ml/topk"])) for u in g["ml/test_u"][:8]])
    assert (full == g["ml/full8"]).mean() > 0.98
