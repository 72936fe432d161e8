"""FM widening on the GPU: the HIP path through the C ABI against the
golden vectors of the REAL reference FM (tests/golden/kat_fm.npz) and the CPU oracle.
Loss within 1e-5 relative, parameters within fp32 round-off, ranked lists identical."""
import os

import numpy as np
import pytest
import torch

from conftest import mf_config
from oracle import fm_numpy as F

pytestmark = pytest.mark.gpu
DEV = "cuda"
HERE = os.path.dirname(os.path.abspath(__file__))
KEYS = ("P", "Q", "bu", "bi", "b0")


@pytest.fixture(scope="module")
def kat_fm():
    return np.load(os.path.join(HERE, "golden", "kat_fm.npz"))


def _t(x):
    return torch.as_tensor(np.ascontiguousarray(x)).to(DEV)


@pytest.mark.parametrize("item_mode", ["sorted", "atomic", "chunked", "fused"])
def test_fm_kat_steps_sgd(kat_fm, item_mode):
    from daisyrec_amd import ops
    g = kat_fm
    for name in g["names"]:
        name = str(name)
        if str(g[f"{name}/optimizer"]) != "sgd":
            continue
        U, I, d, B, ns = (int(x) for x in g[f"{name}/meta"])
        lr, r1, r2 = (float(x) for x in g[f"{name}/hyper"])
        lt = ops.loss_id(str(g[f"{name}/loss_type"]))
        w = [_t(g[f"{name}/{k}0"]) for k in KEYS]
        ctx = ops.BprContext(B, d, U, I)
        ctx.set_pointwise(lt in ops.POINTWISE_LOSSES)
        g_bi = torch.zeros(I, device=DEV)
        ctx.set_bias(w[2], w[3], w[4], g_i_bias=g_bi)
        step_loss = torch.zeros(1, dtype=torch.float64, device=DEV)
        for s in range(ns):
            ctx.set_batch(_t(g[f"{name}/u"][s]), _t(g[f"{name}/i"][s]), _t(g[f"{name}/j"][s]))
            ctx.sgd_step(w[0], w[1], lr, r1, r2, loss_type=lt, item_mode=ops.ITEM_MODES[item_mode],
                         step_loss=step_loss)
            ref = float(g[f"{name}/loss"][s])
            assert abs(float(step_loss.cpu()) - ref) <= 1e-5 * abs(ref), (name, s)
            for k, key in enumerate(KEYS):
                np.testing.assert_allclose(w[k].cpu().numpy().reshape(-1), g[f"{name}/{key}"][s].reshape(-1),
                                           rtol=0, atol=3e-6, err_msg=f"{name} step {s} {key} ({item_mode})")
        assert float(g_bi.abs().max().cpu()) == 0.0 and float(ctx.gQ.abs().max().cpu()) == 0.0
        ctx.close()


@pytest.mark.parametrize("item_mode", ["sorted", "fused"])
def test_fm_kat_adam(kat_fm, item_mode):
    """Adam over the five FM parameters: phase entry points + daisy_adam_dense ('sorted'), and the staged step
    ('fused': the row owners apply torch's Adam to the rows with a gradient, rows without one are replayed by
    flush(); the three bias vectors through the dense optimiser)."""
    from daisyrec_amd import ops
    from daisyrec_amd.model.AbstractRecommender import _AdamState
    g = kat_fm
    for name in ("fm_bpr_adam", "fm_cl_adam"):
        U, I, d, B, ns = (int(x) for x in g[f"{name}/meta"])
        lr, r1, r2 = (float(x) for x in g[f"{name}/hyper"])
        lt = ops.loss_id(str(g[f"{name}/loss_type"]))
        w = [_t(g[f"{name}/{k}0"]) for k in KEYS]
        ctx = ops.BprContext(B, d, U, I)
        ctx.set_pointwise(lt in ops.POINTWISE_LOSSES)
        adam = _AdamState(w[0], w[1], lr, (w[2], w[3], w[4]))
        ctx.set_bias(w[2], w[3], w[4], g_u_bias=adam.g[0], g_i_bias=adam.g[1], g_bias=adam.g[2])
        for s in range(ns):
            ctx.set_batch(_t(g[f"{name}/u"][s]), _t(g[f"{name}/i"][s]), _t(g[f"{name}/j"][s]))
            adam.step(ctx, w[0], w[1], r1, r2, lt, ops.ITEM_MODES[item_mode])
            adam.flush()
            ref = float(g[f"{name}/loss"][s])
            assert abs(float(ctx.stats[7].cpu()) - ref) <= 1e-5 * abs(ref), (name, s)
            for k, key in enumerate(KEYS):
                np.testing.assert_allclose(w[k].cpu().numpy().reshape(-1), g[f"{name}/{key}"][s].reshape(-1),
                                           rtol=0, atol=2e-5, err_msg=f"{name} step {s} {key}")
        ctx.close()


def test_fm_rank_kat(kat_fm):
    from daisyrec_amd import ops
    g = kat_fm
    w = [_t(g[f"rank/{k}"]) for k in KEYS]
    b = (w[2], w[3], w[4])
    topk = int(g["rank/topk"])
    out, scores = ops.mf_rank_topk(w[0], w[1], _t(g["rank/us"]), _t(g["rank/cands"]), topk, return_scores=True,
                                   biases=b)
    np.testing.assert_array_equal(out.cpu().numpy().astype(np.float32), g["rank/preds"])
    _, want = F.fm_rank(*[g[f"rank/{k}"] for k in KEYS], g["rank/us"], g["rank/cands"], topk)
    np.testing.assert_allclose(scores.cpu().numpy(), want, rtol=1e-5, atol=1e-6)
    full = np.stack([ops.mf_full_rank(w[0], w[1], int(u), topk, biases=b).cpu().numpy() for u in g["rank/us"]])
    np.testing.assert_array_equal(full, g["rank/full"])
    pp = ops.mf_predict(w[0], w[1], _t(g["rank/us"]), _t(g["rank/cands"][:, 0]), biases=b)
    np.testing.assert_allclose(pp.cpu().numpy(), g["rank/predict"], rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("B,U,I,d", [(1, 5, 5, 8), (257, 40, 30, 84), (5000, 300, 50, 64), (3000, 3, 2000, 32)])
def test_fm_step_vs_oracle_shapes(B, U, I, d):
    """Long runs (few users / few items) cross the run, slot and chunk-edge paths of the chunked kernels."""
    from daisyrec_amd import ops
    rng = np.random.default_rng(B)
    w0 = [(rng.standard_normal((U, d)) * 0.2).astype(np.float32), (rng.standard_normal((I, d)) * 0.2).astype(np.float32),
          (rng.standard_normal(U) * 0.1).astype(np.float32), (rng.standard_normal(I) * 0.1).astype(np.float32),
          np.array([0.05], np.float32)]
    u, i, j = (rng.integers(0, n, B).astype(np.int32) for n in (U, I, I))
    loss, *want = F.fm_sgd_step(*w0, u, i, j, 0.05, 1e-3, 1e-3)
    for mode in ("sorted", "chunked", "atomic"):
        w = [_t(x) for x in w0]
        ctx = ops.BprContext(B, d, U, I)
        g_bi = torch.zeros(I, device=DEV)
        ctx.set_bias(w[2], w[3], w[4], g_i_bias=g_bi)
        ctx.set_batch(_t(u), _t(i), _t(j))
        sl = torch.zeros(1, dtype=torch.float64, device=DEV)
        ctx.sgd_step(w[0], w[1], 0.05, 1e-3, 1e-3, item_mode=ops.ITEM_MODES[mode], step_loss=sl)
        assert abs(float(sl.cpu()) - loss) <= 1e-5 * abs(loss)
        for k, key in enumerate(KEYS):
            np.testing.assert_allclose(w[k].cpu().numpy().reshape(-1), np.asarray(want[k]).reshape(-1), rtol=0,
                                       atol=2e-5 if k >= 2 else 5e-6, err_msg=f"{key} ({mode})")
        ctx.close()


def test_fm_ml100k_through_the_dropin(kat_fm):
    """run_examples/test.py --algo_name fm on ml-100k (fm.yaml: d=84, lr 0.001, SGD, B=256) through
    FM.fit / FM.rank with the reference's triples, init and DataLoader order."""
    from daisyrec_amd.model.FMRecommender import FM
    from daisyrec_amd.utils.dataset import BasicDataset, CandidatesDataset, get_dataloader
    g = kat_fm
    lr, r1, r2 = (float(x) for x in g["ml/hyper"])
    cfg = mf_config(user_num=int(g["ml/user_num"]), item_num=int(g["ml/item_num"]), epochs=int(g["ml/epochs"]),
                    factors=int(g["ml/factors"]), lr=lr, reg_1=r1, reg_2=r2, algo_name="fm")
    torch.manual_seed(int(g["ml/seed"]))
    model = FM(cfg)
    np.testing.assert_array_equal(model.embed_user.weight.detach().numpy(), g["ml/P0"])
    np.testing.assert_array_equal(model.embed_item.weight.detach().numpy(), g["ml/Q0"])
    assert float(model.u_bias.weight.detach().abs().max()) == 0.0 and float(model.bias_.detach()) == 0.0
    loader = get_dataloader(BasicDataset(g["ml/samples"]), batch_size=int(g["ml/batch_size"]), shuffle=True,
                            num_workers=4)
    torch.set_rng_state(torch.from_numpy(g["ml/rng_state_before_fit"]))
    model.fit(loader)
    for got, ref in zip(model.epoch_losses, g["ml/epoch_losses"]):
        assert abs(got - ref) <= 1e-5 * abs(ref), (got, ref)
    final = [model.embed_user.weight, model.embed_item.weight, model.u_bias.weight, model.i_bias.weight, model.bias_]
    for p, key in zip(final, KEYS):
        np.testing.assert_allclose(p.detach().cpu().numpy().reshape(-1), g[f"ml/{key}1"].reshape(-1), atol=2e-4)
    ucands = [[int(u), c] for u, c in zip(g["ml/test_u"], g["ml/cands"])]
    preds = model.rank(get_dataloader(CandidatesDataset(ucands), batch_size=128, shuffle=False, num_workers=0))
    assert preds.dtype == np.float32 and preds.shape == g["ml/preds"].shape
    np.testing.assert_array_equal(preds, g["ml/preds"])
    want = F.fm_forward(*[g[f"ml/{k}1"] for k in KEYS], [3], [5])[0]
    assert abs(model.predict(3, 5) - float(want)) < 1e-4
    # throughput mode and Adam run and converge too
    for kw in (dict(item_mode="chunked"), dict(optimizer="adam", lr=0.001)):
        torch.manual_seed(1)
        m2 = FM({**cfg, **kw, "epochs": 2})
        m2.fit(loader)
        assert m2.epoch_losses[1] < m2.epoch_losses[0]
        assert float(m2.i_bias.weight.abs().max()) > 0.0
