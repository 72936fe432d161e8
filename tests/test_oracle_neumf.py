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


def _replay_d64(bf16_points=None):
    """tests/golden/kat_neumf_d64.npz (make_golden_neumf_d64.py: the REFERENCE's fit at the configs[3] tower shape) replayed
    with the numpy oracle: the model's init from the seed (checked against the reference's checksums), the DataLoader's
    order from the stored RNG state, 12 SGD steps"""
    import logging
    from daisyrec_amd.model.NeuMFRecommender import NeuMF
    g = np.load(os.path.join(HERE, "golden", "kat_neumf_d64.npz"))
    U, I, d, L = (int(x) for x in g["meta"])
    lr, r1, r2 = (float(x) for x in g["hyper"])
    cfg = {"gpu": "0", "logger": logging.getLogger("t"), "lr": lr, "reg_1": r1, "reg_2": r2, "epochs": 1, "topk": 50,
           "user_num": U, "item_num": I, "factors": d, "num_layers": L, "dropout": 0.0, "loss_type": "BPR",
           "optimizer": "sgd", "init_method": "default", "early_stop": False, "model_name": "NeuMF", "GMF_model": None,
           "MLP_model": None, "algo_name": "neumf", "progress": False}
    torch.manual_seed(int(g["seed"]))
    model = NeuMF(cfg)
    names = N.param_names(L)
    p = {k: v.detach().cpu().numpy().copy() for k, v in model._named().items()}
    for k in names:
        np.testing.assert_allclose([p[k].astype(np.float64).sum(), np.abs(p[k].astype(np.float64)).sum()], g[f"{k}0_sum"], rtol=1e-12)
    init = {k: v.copy() for k, v in p.items()}
    samples, B = g["samples"], int(g["batch_size"])
    n = len(samples)
    torch.set_rng_state(torch.from_numpy(g["rng_state_before_fit"]))
    torch.empty((), dtype=torch.int64).random_()
    gen = torch.Generator()
    gen.manual_seed(int(torch.empty((), dtype=torch.int64).random_().item()))
    perm = torch.randperm(n, generator=gen).numpy()
    tot = 0.0
    for s in range(0, n, B):
        idx = perm[s:s + B]
        loss, grads = N.neumf_grad(p, samples[idx, 0], samples[idx, 1], samples[idx, 2], r1, r2, L, bf16_points=bf16_points)
        p = {k: (np.asarray(p[k], np.float64) - lr * grads[k]).astype(np.float32) for k in names}
        tot += loss
    return g, p, init, tot


def test_neumf_d64_golden_fp64_oracle():
    """the oracle against the reference's fit at factors 64 / 3 layers: epoch loss within 1e-5, the 12-step change of every
    parameter at round-off"""
    g, p, init, tot = _replay_d64()
    ref = float(g["epoch_losses"][0])
    assert abs(tot - ref) <= 1e-5 * abs(ref), (tot, ref)
    for k in p:
        rows = g[f"{k}_delta"].shape[0]
        want = g[f"{k}_delta"]
        got = (p[k] - init[k])[:rows]
        assert np.linalg.norm(got - want) <= 2e-3 * np.linalg.norm(want) + 1e-9, k
        np.testing.assert_allclose(p[k][:rows], g[f"{k}1"], atol=2e-6, err_msg=k)


def test_neumf_d64_golden_bf16_oracle_states_the_distance_of_bf16_arithmetic():
    """the SAME fit with the roundings of the HIP path's bf16-storage mode (neumf_grad_bf16 'fact'): how far bf16 arithmetic
    ITSELF is from the reference over twelve steps.  Measured here: the epoch loss 2.8e-6 away, the GMF tables' change 7e-5
    (that branch never leaves fp32), the MLP parameters' change 2.3 ... 3.7 % - a BPR step's MLP gradient is the DIFFERENCE of the
    positive and the negative row's nearly equal contributions, so a 2^-9 rounding of either side is a few per cent of what
    is left.  The GPU test holds the bf16 fit to the reference with these distances plus a margin, and to THIS replay tightly
    (tests/test_gpu_neumf.py::test_neumf_ml100k_d64_fit_against_the_reference)."""
    g, p, init, tot = _replay_d64("fact")
    ref = float(g["epoch_losses"][0])
    assert 1e-7 < abs(tot - ref) / abs(ref) <= 5e-5, (tot, ref)
    err = {}
    for k in p:
        rows = g[f"{k}_delta"].shape[0]
        want = g[f"{k}_delta"]
        err[k] = float(np.linalg.norm((p[k] - init[k])[:rows] - want) / max(np.linalg.norm(want), 1e-30))
    assert err["uG"] <= 1e-3 and err["iG"] <= 1e-3, err
    assert all(e <= 0.06 for e in err.values()) and max(err.values()) > 0.01, err
