"""oracle/neumf_numpy.py against the golden vectors the REAL reference NeuMF produced
(tests/golden/make_golden_neumf.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import bpr_mf_numpy as O
from oracle import neumf_numpy as N

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def kat_neumf():
    return np.load(os.path.join(HERE, "golden", "kat_neumf.npz"))


def assert_params_close(got, ref, names, tag, atol, adam_lr=None, steps=1, frac=0.999):
    """SGD: every element within atol.  Adam: m/sqrt(v) is sign-like where a gradient is ~0, so a
    last-ulp difference in such a gradient moves the weight by up to lr per step: require `frac`
    of the elements within atol and all of them within 2*lr*steps."""
    for k in names:
        a, b = np.asarray(got[k], np.float64).reshape(-1), np.asarray(ref[k], np.float64).reshape(-1)
        diff = np.abs(a - b)
        if adam_lr is None:
            assert diff.max() <= atol, f"{tag} {k}: max diff {diff.max():.3e}"
        else:
            assert (diff <= atol).mean() >= frac and diff.max() <= 2 * adam_lr * steps + atol, \
                f"{tag} {k}: {(diff > atol).sum()} of {diff.size} beyond {atol}, max {diff.max():.3e}"


def load_params(g, prefix, L, suffix=""):
    return {k: g[f"{prefix}/{k}{suffix}"] for k in N.param_names(L)}


def oracle_steps(g, name, dtype=np.float64):
    U, I, d, L, B, ns = (int(x) for x in g[f"{name}/meta"])
    lr, r1, r2 = (float(x) for x in g[f"{name}/hyper"])
    lt = O.LOSS_IDS[str(g[f"{name}/loss_type"])]
    model = str(g[f"{name}/model"])
    p = load_params(g, name, L, "0")
    names = N.param_names(L)
    adam = O.DenseAdam([p[k].shape for k in names], lr, dtype=dtype) if str(g[f"{name}/optimizer"]) == "adam" else None
    for s in range(ns):
        loss, grads = N.neumf_grad(p, g[f"{name}/u"][s], g[f"{name}/i"][s], g[f"{name}/j"][s], r1, r2, L, lt,
                                   model, dtype=dtype)
        if adam is None:
            p = {k: (np.asarray(p[k], dtype) - lr * grads[k]).astype(np.float32) for k in names}
        else:
            p = dict(zip(names, adam.step([p[k] for k in names], [grads[k] for k in names])))
        yield s, loss, p


def test_neumf_kat_steps(kat_neumf):
    g = kat_neumf
    for name in g["names"]:
        name = str(name)
        L = int(g[f"{name}/meta"][3])
        for s, loss, p in oracle_steps(g, name):
            ref = g[f"{name}/loss"][s]
            assert abs(loss - ref) <= 5e-6 * abs(ref), (name, s, loss, ref)
        is_adam = str(g[f"{name}/optimizer"]) == "adam"
        assert_params_close(p, {k: g[f"{name}/{k}"] for k in N.param_names(L)}, N.param_names(L), name, 3e-6,
                            adam_lr=float(g[f"{name}/hyper"][0]) if is_adam else None,
                            steps=int(g[f"{name}/meta"][5]))


def test_neumf_rank_kat(kat_neumf):
    g = kat_neumf
    U, I, d, L = (int(x) for x in g["rank/meta"])
    p = load_params(g, "rank", L)
    pred, _ = N.neumf_rank(p, g["rank/us"], g["rank/cands"], int(g["rank/topk"]), L)
    assert (pred == g["rank/preds"]).mean() > 0.99      # fp32 sum order of torch's GEMMs differs in the last ulp
    full = np.stack([N.neumf_full_rank(p, int(u), int(g["rank/topk"]), L) for u in g["rank/us"]])
    assert (full == g["rank/full"]).mean() > 0.99
    pp, _ = N.neumf_forward(p, g["rank/us"], g["rank/cands"][:, 0], L)
    np.testing.assert_allclose(pp, g["rank/predict"], rtol=1e-5, atol=1e-6)


def _replay_ml100k(g, prefix):
    U, I, d, L = (int(x) for x in g[f"{prefix}/meta"])
    samples, B = g["ml/samples"], int(g[f"{prefix}/batch_size"])
    lr, r1, r2 = (float(x) for x in g[f"{prefix}/hyper"])
    names = N.param_names(L)
    p = load_params(g, prefix, L, "0")
    adam = O.DenseAdam([p[k].shape for k in names], lr) if str(g[f"{prefix}/optimizer"]) == "adam" else None
    n = len(samples)
    torch.set_rng_state(torch.from_numpy(g[f"{prefix}/rng_state_before_fit"]))
    losses = []
    for ep in range(int(g[f"{prefix}/epochs"])):
        torch.empty((), dtype=torch.int64).random_()
        gen = torch.Generator()
        gen.manual_seed(int(torch.empty((), dtype=torch.int64).random_().item()))
        perm = torch.randperm(n, generator=gen).numpy()
        tot = 0.0
        for s in range(0, n, B):
            idx = perm[s:s + B]
            loss, grads = N.neumf_grad(p, samples[idx, 0], samples[idx, 1], samples[idx, 2], r1, r2, L)
            if adam is None:
                p = {k: (np.asarray(p[k], np.float64) - lr * grads[k]).astype(np.float32) for k in names}
            else:
                p = dict(zip(names, adam.step([p[k] for k in names], [grads[k] for k in names])))
            tot += loss
        losses.append(tot)
    return p, losses, L


def test_neumf_ml100k_sgd_end_to_end(kat_neumf):
    """run_examples/test.py --algo_name neumf --optimizer sgd (dropout 0): epoch loss within 1e-5 and
    identical top-N (SGD is smooth, so the lists survive round-off)."""
    g = kat_neumf
    p, losses, L = _replay_ml100k(g, "mlsgd")
    ref = g["mlsgd/epoch_losses"][0]
    assert abs(losses[0] - ref) <= 1e-5 * abs(ref)
    for k in N.param_names(L):
        np.testing.assert_allclose(p[k], g[f"mlsgd/{k}1"], atol=2e-5, err_msg=k)
    pred, _ = N.neumf_rank(p, g["ml/test_u"], g["ml/cands"], int(g["mlsgd/topk"]), L)
    same = (pred == g["mlsgd/preds"]).all(axis=1).mean()
    assert same > 0.98, f"top-N lists identical for {same:.3f} of the users"


def test_neumf_ml100k_adam_end_to_end(kat_neumf):
    """neumf.yaml defaults (Adam, lr 0.001; dropout 0), 100 batches."""
    g = kat_neumf
    p, losses, L = _replay_ml100k(g, "ml")
    ref = g["ml/epoch_losses"][0]
    assert abs(losses[0] - ref) <= 1e-5 * abs(ref), (losses, ref)
    for k in N.param_names(L):
        ref = g[f"ml/{k}1"].astype(np.float64)
        err = np.linalg.norm(np.asarray(p[k], np.float64) - ref)
        assert err <= 1e-3 * np.linalg.norm(ref) + 1e-6, (k, err)      # (bp stays exactly 0 under BPR)
    pred, _ = N.neumf_rank(p, g["ml/test_u"], g["ml/cands"], int(g["ml/topk"]), L)
    same = (pred == g["ml/preds"]).all(axis=1).mean()
    assert same > 0.9, f"top-N lists identical for {same:.3f} of the users"
