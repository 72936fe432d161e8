"""oracle/fm_numpy.py against the golden vectors the REAL reference FM produced
(tests/golden/make_golden_fm.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import bpr_mf_numpy as O
from oracle import fm_numpy as F

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def kat_fm():
    return np.load(os.path.join(HERE, "golden", "kat_fm.npz"))


KEYS = ("P", "Q", "bu", "bi", "b0")


def run_fm_case(g, name, dtype):
    lr, r1, r2 = g[f"{name}/hyper"]
    lt = O.LOSS_IDS[str(g[f"{name}/loss_type"])]
    w = [g[f"{name}/{k}0"] for k in KEYS]
    w[2], w[3] = w[2].reshape(-1), w[3].reshape(-1)
    adam = O.DenseAdam([x.shape for x in w], lr, dtype=dtype) if str(g[f"{name}/optimizer"]) == "adam" else None
    for s in range(int(g[f"{name}/meta"][4])):
        u, i, j = g[f"{name}/u"][s], g[f"{name}/i"][s], g[f"{name}/j"][s]
        if adam is None:
            loss, *w = F.fm_sgd_step(*w, u, i, j, lr, r1, r2, lt, dtype=dtype)
            w[4] = np.array([w[4]], dtype=np.float32)
        else:
            loss, *grads = F.fm_grad(*w, u, i, j, r1, r2, lt, dtype=dtype)
            grads[4] = np.array([grads[4]], dtype=dtype)
            w = adam.step(w, grads)
        yield s, loss, w


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_fm_kat_steps(kat_fm, dtype):
    g = kat_fm
    for name in g["names"]:
        name = str(name)
        for s, loss, w in run_fm_case(g, name, dtype):
            ref = g[f"{name}/loss"][s]
            assert abs(loss - ref) <= 3e-6 * abs(ref), (name, s, loss, ref)
            for k, key in enumerate(KEYS):
                np.testing.assert_allclose(np.asarray(w[k]).reshape(-1), g[f"{name}/{key}"][s].reshape(-1),
                                           rtol=0, atol=2e-6, err_msg=f"{name} step {s} {key}")


def test_fm_rank_kat(kat_fm):
    g = kat_fm
    w = [g[f"rank/{k}"] for k in KEYS]
    pred, _ = F.fm_rank(*w, g["rank/us"], g["rank/cands"], int(g["rank/topk"]))
    np.testing.assert_array_equal(pred, g["rank/preds"])
    full = np.stack([F.fm_full_rank(*w, int(u), int(g["rank/topk"])) for u in g["rank/us"]])
    np.testing.assert_array_equal(full, g["rank/full"])
    pp = F.fm_forward(*w, g["rank/us"], g["rank/cands"][:, 0])
    np.testing.assert_allclose(pp, g["rank/predict"], rtol=1e-5, atol=1e-7)


def fm_epoch_orders(g):
    n = len(g["ml/samples"])
    torch.set_rng_state(torch.from_numpy(g["ml/rng_state_before_fit"]))
    for _ in range(int(g["ml/epochs"])):
        torch.empty((), dtype=torch.int64).random_()
        seed = int(torch.empty((), dtype=torch.int64).random_().item())
        gen = torch.Generator()
        gen.manual_seed(seed)
        yield torch.randperm(n, generator=gen).numpy()


def test_fm_ml100k_end_to_end(kat_fm):
    """run_examples/test.py --algo_name fm on ml-100k: epoch losses within 1e-5, identical top-N."""
    g = kat_fm
    samples, B = g["ml/samples"], int(g["ml/batch_size"])
    lr, r1, r2 = g["ml/hyper"]
    w = [g[f"ml/{k}0"].copy() for k in KEYS]
    w[2], w[3] = w[2].reshape(-1), w[3].reshape(-1)
    for ep, perm in enumerate(fm_epoch_orders(g)):
        tot = 0.0
        for s in range(0, len(samples), B):
            idx = perm[s:s + B]
            loss, *w = F.fm_sgd_step(*w, samples[idx, 0], samples[idx, 1], samples[idx, 2], lr, r1, r2)
            tot += loss
        ref = g["ml/epoch_losses"][ep]
        assert abs(tot - ref) <= 1e-5 * abs(ref)
    for k, key in enumerate(KEYS):
        np.testing.assert_allclose(np.asarray(w[k]).reshape(-1), g[f"ml/{key}1"].reshape(-1), atol=2e-4)
    pred, scores = F.fm_rank(*w, g["ml/test_u"], g["ml/cands"], int(g["ml/topk"]))
    same = (pred == g["ml/preds"]).all(axis=1).mean()
    assert same == 1.0, f"top-N lists identical for {same:.3f} of the users"
