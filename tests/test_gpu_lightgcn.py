"""LightGCN widening on the GPU: the HIP path through the C ABI against the golden
vectors of the REAL reference LightGCN (tests/golden/kat_lightgcn.npz) and the CPU oracle."""
import os

import numpy as np
import pytest
import scipy.sparse as sp
import torch

from conftest import mf_config
from oracle import lightgcn_numpy as LG
from test_oracle_neumf import assert_params_close

pytestmark = pytest.mark.gpu
DEV = "cuda"
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def kat_lg():
    return np.load(os.path.join(HERE, "golden", "kat_lightgcn.npz"))


def _t(x):
    return torch.as_tensor(np.ascontiguousarray(x)).to(DEV)


def test_graph_is_the_reference_adjacency_and_spmm(kat_lg):
    from daisyrec_amd import ops
    g = kat_lg
    U, I, d, L = (int(x) for x in g["rank/meta"])
    graph = ops.LgcnGraph(_t(g["rank/gu"]), _t(g["rank/gi"]), U, I)
    row, col, val = (t.cpu().numpy() for t in graph.coo())
    order = np.lexsort((g["rank/adj_col"], g["rank/adj_row"]))
    np.testing.assert_array_equal(row, g["rank/adj_row"][order])           # integer work: bit exact
    np.testing.assert_array_equal(col, g["rank/adj_col"][order])
    ulp = np.abs(val.view(np.int32) - g["rank/adj_val"][order].view(np.int32))
    assert ulp.max() <= 1                                                  # float64 pow on device vs numpy: <= 1 ulp of fp32
    X = np.random.default_rng(0).standard_normal((U + I, d)).astype(np.float32)
    want = LG.spmm(LG.norm_adj_csr(g["rank/gu"], g["rank/gi"], U, I), X.astype(np.float64))
    np.testing.assert_allclose(graph.spmm(_t(X)).cpu().numpy(), want, rtol=1e-5, atol=1e-6)
    out = graph.propagate(_t(X), 3).cpu().numpy()
    np.testing.assert_allclose(out, LG.propagate(LG.norm_adj_csr(g["rank/gu"], g["rank/gi"], U, I), X.astype(np.float64), 3),
                               rtol=1e-5, atol=1e-6)
    graph.close()


@pytest.mark.parametrize("U,I,n,d", [(3, 2, 7, 8), (2000, 10, 30000, 64), (50, 5000, 20001, 32), (300, 300, 1, 20)])
def test_spmm_shapes_long_and_short_rows(U, I, n, d):
    """few items -> item rows with thousands of neighbours (chunk-crossing segments), isolated nodes, d % 4 != 0"""
    from daisyrec_amd import ops
    rng = np.random.default_rng(n)
    gu, gi = rng.integers(0, U, n), rng.integers(0, I, n)
    graph = ops.LgcnGraph(_t(gu), _t(gi), U, I)
    X = rng.standard_normal((U + I, d)).astype(np.float32)
    want = LG.spmm(LG.norm_adj_csr(gu, gi, U, I), X.astype(np.float64))
    np.testing.assert_allclose(graph.spmm(_t(X)).cpu().numpy(), want, rtol=2e-5, atol=2e-6)
    graph.close()


def _model(g, prefix, **over):
    from daisyrec_amd.model.LightGCNRecommender import LightGCN
    U, I, d, L = (int(x) for x in g[f"{prefix}/meta"][:4])
    gu, gi = (g[f"{prefix}/gu"], g[f"{prefix}/gi"]) if f"{prefix}/gu" in g else (g["ml/train_users"], g["ml/train_items"])
    cfg = mf_config(user_num=U, item_num=I, factors=d, num_layers=L, algo_name="lightgcn", reg_1=0.0, reg_2=0.0, lr=0.01,
                    inter_matrix=sp.coo_matrix((np.ones(len(gu), np.float32), (gu, gi)), shape=(U, I)))
    cfg.update(over)
    return LightGCN(cfg), L


def test_lightgcn_kat_steps(kat_lg):
    """LightGCN.calc_loss -> backward -> optimiser step, step by step against the reference."""
    from daisyrec_amd import ops
    g = kat_lg
    for name in g["names"]:
        name = str(name)
        U, I, d, L, B, ns = (int(x) for x in g[f"{name}/meta"])
        lr, r1, r2 = (float(x) for x in g[f"{name}/hyper"])
        lt = str(g[f"{name}/loss_type"])
        model, _ = _model(g, name, reg_1=r1, reg_2=r2, lr=lr, loss_type=lt, optimizer=str(g[f"{name}/optimizer"]))
        with torch.no_grad():
            model.embed_user.weight.copy_(torch.from_numpy(g[f"{name}/P0"]))
            model.embed_item.weight.copy_(torch.from_numpy(g[f"{name}/Q0"]))
        E0 = model._ego()
        loss_id = ops.loss_id(lt)
        ctx = ops.BprContext(B, d, U, I)
        ctx.set_pointwise(loss_id in ops.POINTWISE_LOSSES)
        out, G, dE0 = torch.empty_like(E0), torch.empty_like(E0), torch.zeros_like(E0)
        is_adam = model.optimizer == "adam"
        m, v = torch.zeros_like(model._flat), torch.zeros_like(model._flat)
        for s in range(ns):
            model._batch_grads(ctx, E0, out, G, dE0, _t(g[f"{name}/u"][s]), _t(g[f"{name}/i"][s]), _t(g[f"{name}/j"][s]), loss_id)
            loss = float(ctx.stats[7].cpu())
            ref = float(g[f"{name}/loss"][s])
            assert abs(loss - ref) <= 1e-5 * abs(ref), (name, s, loss, ref)
            if is_adam:
                ops.adam_dense(model._flat, dE0.view(-1), m, v, lr, s + 1)
            else:
                ops.sgd_dense(model._flat, dE0.view(-1), lr)
        got = {"P": model.embed_user.weight.detach().cpu().numpy(), "Q": model.embed_item.weight.detach().cpu().numpy()}
        assert_params_close(got, {"P": g[f"{name}/P"], "Q": g[f"{name}/Q"]}, ("P", "Q"), name, 5e-6,
                            adam_lr=lr if is_adam else None, steps=ns, frac=0.98)
        ctx.close()


def test_lightgcn_rank_kat(kat_lg):
    from daisyrec_amd.utils.dataset import CandidatesDataset, get_dataloader
    g = kat_lg
    model, L = _model(g, "rank", topk=int(g["rank/topk"]))
    with torch.no_grad():
        model.embed_user.weight.copy_(torch.from_numpy(g["rank/P"]))
        model.embed_item.weight.copy_(torch.from_numpy(g["rank/Q"]))
    ucands = [[int(u), c] for u, c in zip(g["rank/us"], g["rank/cands"])]
    preds = model.rank(get_dataloader(CandidatesDataset(ucands), batch_size=4, shuffle=False, num_workers=0))
    assert preds.dtype == np.float32 and (preds == g["rank/preds"]).mean() > 0.97
    full = np.stack([model.full_rank(int(u)) for u in g["rank/us"]])
    assert (full == g["rank/full"]).mean() > 0.97
    pp = np.array([model.predict(int(u), int(c[0])) for u, c in zip(g["rank/us"], g["rank/cands"])])
    np.testing.assert_allclose(pp, g["rank/predict"], rtol=1e-5, atol=1e-6)


def test_lightgcn_ml100k_through_the_dropin(kat_lg):
    """run_examples/test.py --algo_name lightgcn on ml-100k (lightgcn.yaml: d=64, 2 layers, Adam lr 0.01; first 50
    batches) through LightGCN.fit / rank with the reference's graph, triples, init and DataLoader order."""
    from daisyrec_amd.utils.dataset import BasicDataset, CandidatesDataset, get_dataloader
    g = kat_lg
    torch.manual_seed(int(g["ml/seed"]))
    model, L = _model(g, "ml", epochs=1, topk=int(g["ml/topk"]))
    np.testing.assert_array_equal(model.embed_user.weight.detach().numpy(), g["ml/P0"])
    np.testing.assert_array_equal(model.embed_item.weight.detach().numpy(), g["ml/Q0"])
    loader = get_dataloader(BasicDataset(g["ml/samples"]), batch_size=int(g["ml/batch_size"]), shuffle=True, num_workers=4)
    torch.set_rng_state(torch.from_numpy(g["ml/rng_state_before_fit"]))
    model.fit(loader)
    ref = float(g["ml/epoch_losses"][0])
    assert abs(model.epoch_losses[0] - ref) <= 1e-5 * abs(ref), (model.epoch_losses, ref)
    for got, key in ((model.embed_user.weight, "P1"), (model.embed_item.weight, "Q1")):
        a = got.detach().cpu().numpy().astype(np.float64)
        assert np.linalg.norm(a - g[f"ml/{key}"]) / np.linalg.norm(g[f"ml/{key}"]) < 1e-3
    ucands = [[int(u), c] for u, c in zip(g["ml/test_u"], g["ml/cands"])]
    preds = model.rank(get_dataloader(CandidatesDataset(ucands), batch_size=128, shuffle=False, num_workers=0))
    same = (preds == g["ml/preds"]).all(axis=1).mean()
    assert same > 0.9, f"top-N lists identical for {same:.3f} of the users"
    b = g["ml/samples"][:256]
    loss = float(model.calc_loss([torch.from_numpy(b[:, k].copy()) for k in range(3)]).cpu())
    graph = LG.norm_adj_csr(g["ml/train_users"], g["ml/train_items"], model.user_num, model.item_num)
    want, _, _ = LG.lightgcn_grad(graph, model.embed_user.weight.detach().cpu().numpy(),
                                  model.embed_item.weight.detach().cpu().numpy(), b[:, 0], b[:, 1], b[:, 2], 0.0, 0.0, L)
    assert abs(loss - want) <= 1e-5 * abs(want)


def test_lightgcn_argument_errors():
    from daisyrec_amd import ops
    u = torch.zeros(4, dtype=torch.int64, device=DEV)
    with pytest.raises(ValueError):
        ops.LgcnGraph(u, u, 0, 5)                               # no users
    with pytest.raises((RuntimeError, ValueError)):
        ops.LgcnGraph(u.cpu(), u.cpu(), 5, 5)                   # host tensors: no CPU fallback
    g = ops.LgcnGraph(u, u, 5, 5)
    assert g.nnz == 2                                           # four copies of one interaction collapse
    X = torch.zeros(10, 8, device=DEV)
    with pytest.raises(ValueError):
        ops.check(ops.lib.daisy_lgcn_spmm(g._h, ops._ptr(X, torch.float32, "X"), ops._ptr(X, torch.float32, "X"), 8, None))
    g.close()


def test_lightgcn_reproducible_mode_is_bitwise_repeatable(kat_lg):
    """Two runs of the same training give identical bits in both modes, on a graph whose item rows span many
    chunks ('sorted' = row-owner products, 'chunked' = the default segmented reduction with parked partials
    and edge records); the two modes agree to round-off."""
    from daisyrec_amd.model.LightGCNRecommender import LightGCN
    from daisyrec_amd.utils.dataset import BasicDataset, get_dataloader
    rng = np.random.default_rng(1)
    U, I, d, n = 3000, 40, 64, 60000                         # 40 items x ~1500 users each: long rows
    gu, gi = rng.integers(0, U, n), rng.integers(0, I, n)
    samples = np.stack([gu, gi, rng.integers(0, I, n)], 1).astype(np.int32)[:8192]
    outs = {}
    for mode in ("sorted", "sorted", "chunked", "chunked"):
        torch.manual_seed(0)
        cfg = mf_config(user_num=U, item_num=I, factors=d, num_layers=2, algo_name="lightgcn", reg_1=0.0, reg_2=0.0, lr=0.01,
                        epochs=1, item_mode=mode, batch_size=1024,
                        inter_matrix=sp.coo_matrix((np.ones(n, np.float32), (gu, gi)), shape=(U, I)))
        model = LightGCN(cfg)
        model.fit(get_dataloader(BasicDataset(samples), batch_size=1024, shuffle=False, num_workers=0))
        outs.setdefault(mode, []).append((model.epoch_losses[0], model.embed_user.weight.detach().cpu().numpy().copy(),
                                          model.embed_item.weight.detach().cpu().numpy().copy()))
    (l0, p0, q0), (l1, p1, q1) = outs["sorted"]
    assert l0 == l1 and np.array_equal(p0, p1) and np.array_equal(q0, q1)
    lc, pc, qc = outs["chunked"][0]
    lc1, pc1, qc1 = outs["chunked"][1]                       # the default kernels are reproducible too
    assert lc == lc1 and np.array_equal(pc, pc1) and np.array_equal(qc, qc1)
    assert abs(lc - l0) <= 1e-6 * abs(l0)
    assert np.abs(pc - p0).max() < 0.05 and np.abs(qc - q0).max() < 0.05      # Adam: bounded by a few lr
