#!/usr/bin/env python
"""Golden vectors for the LightGCN widening, generated from the REAL reference
(`daisy.model.LightGCNRecommender.LightGCN`, imported from /root/reference; nothing is copied).  Runs
only in the build container; the output tests/golden/kat_lightgcn.npz is committed.

    python tests/golden/make_golden_lightgcn.py

  (1) the normalised adjacency of a small random interaction set with duplicate pairs
      (LightGCNRecommender.py:74-107), as COO (row, col, float32 value);
  (2) step KATs through LightGCN.calc_loss -> backward -> optimizer.step (:117-169): BPR/TL/CL, Adam (the
      model's default) and SGD, 1-3 layers, with and without regularisers;
  (3) rank / full_rank / predict (:171-210);
  (4) ml-100k in run_examples/test.py's call order with --algo_name lightgcn (lightgcn.yaml: d=64,
      2 layers, lr 0.01, Adam, reg 0): one epoch over the first 12 800 triples (50 batches).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G  # noqa: E402

import scipy.sparse as sp  # noqa: E402
import torch  # noqa: E402

if not hasattr(sp.dok_matrix, "_update"):     # LightGCNRecommender.py:89 calls a private scipy method that
    sp.dok_matrix._update = lambda self, data: self._dict.update(data)   # scipy >= 1.13 removed (same effect)
import yaml  # noqa: E402
from daisy.model.LightGCNRecommender import LightGCN  # noqa: E402
import daisy.model.AbstractRecommender as ref_abs  # noqa: E402
from daisy.utils.dataset import BasicDataset, CandidatesDataset, get_dataloader  # noqa: E402
from daisy.utils.loader import Preprocessor, RawDataReader  # noqa: E402
from daisy.utils.sampler import BasicNegtiveSampler  # noqa: E402
from daisy.utils.splitter import TestSplitter  # noqa: E402
from daisy.utils.utils import build_candidates_set, get_inter_matrix, get_ur  # noqa: E402


def lg_config(**over):
    cfg = G.base_config()
    cfg.update(yaml.safe_load(open(os.path.join(G.REF, "daisy/assets/lightgcn.yaml"))))
    cfg.update(over)
    return cfg


def random_graph(rng, U, I, n):
    gu, gi = rng.integers(0, U, n), rng.integers(0, I, n)
    gu[:5], gi[:5] = gu[5:10], gi[5:10]                      # duplicate interactions
    return gu.astype(np.int64), gi.astype(np.int64)


def make_model(cfg, gu, gi):
    cfg["inter_matrix"] = sp.coo_matrix((np.ones(len(gu), np.float32), (gu, gi)),
                                        shape=(cfg["user_num"], cfg["item_num"]))
    return LightGCN(cfg)


def kat_case(name, U, I, d, L, nedge, B, loss_type, optimizer, reg, lr, n_steps, rng):
    gu, gi = random_graph(rng, U, I, nedge)
    cfg = lg_config(user_num=U, item_num=I, factors=d, num_layers=L, loss_type=loss_type, optimizer=optimizer,
                    reg_1=reg, reg_2=reg, lr=lr, epochs=1, early_stop=False, init_method="default")
    torch.manual_seed(int(rng.integers(1 << 30)))
    model = make_model(cfg, gu, gi)
    P0, Q0 = model.embed_user.weight.detach().numpy().copy(), model.embed_item.weight.detach().numpy().copy()
    opt = model._build_optimizer(optimizer=model.optimizer, lr=model.lr)
    model.criterion = model._build_criterion(model.loss_type)
    us, is_, js, losses = [], [], [], []
    for _ in range(n_steps):
        u = rng.integers(0, U, size=B).astype(np.int32)
        i = rng.integers(0, I, size=B).astype(np.int32)
        j = (rng.integers(0, 2, size=B) if loss_type in ("CL", "SL") else rng.integers(0, I, size=B)).astype(np.int32)
        u[1] = u[0]; i[2] = i[0]
        if loss_type not in ("CL", "SL"):
            j[3] = i[0]
        model.zero_grad()
        loss = model.calc_loss([torch.from_numpy(x) for x in (u, i, j)])
        loss.backward()
        opt.step()
        us.append(u); is_.append(i); js.append(j)
        losses.append(float(loss.item()))
    return {f"{name}/meta": np.array([U, I, d, L, B, n_steps], dtype=np.int64),
            f"{name}/hyper": np.array([lr, reg, reg], dtype=np.float64),
            f"{name}/loss_type": np.array(loss_type), f"{name}/optimizer": np.array(model.optimizer),
            f"{name}/gu": gu, f"{name}/gi": gi, f"{name}/P0": P0, f"{name}/Q0": Q0,
            f"{name}/u": np.stack(us), f"{name}/i": np.stack(is_), f"{name}/j": np.stack(js),
            f"{name}/loss": np.array(losses, dtype=np.float64),
            f"{name}/P": model.embed_user.weight.detach().numpy().copy(),
            f"{name}/Q": model.embed_item.weight.detach().numpy().copy()}


def adj_and_rank_case(rng):
    U, I, d, L, C, nB, topk = 40, 60, 16, 2, 30, 10, 10
    gu, gi = random_graph(rng, U, I, 400)
    torch.manual_seed(3)
    model = make_model(lg_config(user_num=U, item_num=I, factors=d, num_layers=L, topk=topk), gu, gi)
    A = model.norm_adj_matrix.coalesce()
    us = rng.integers(0, U, size=nB).astype(np.int64)
    cands = rng.integers(0, I, size=(nB, C)).astype(np.int64)
    loader = get_dataloader(CandidatesDataset([[int(us[b]), cands[b]] for b in range(nB)]), batch_size=4,
                            shuffle=False, num_workers=0)
    preds = model.rank(loader)
    full = np.stack([model.full_rank(int(u)) for u in us])
    pred_pairs = np.array([model.predict(int(us[b]), int(cands[b, 0])) for b in range(nB)], dtype=np.float32)
    return {"rank/meta": np.array([U, I, d, L], dtype=np.int64), "rank/gu": gu, "rank/gi": gi,
            "rank/adj_row": A.indices()[0].numpy(), "rank/adj_col": A.indices()[1].numpy(),
            "rank/adj_val": A.values().numpy(),
            "rank/P": model.embed_user.weight.detach().numpy().copy(),
            "rank/Q": model.embed_item.weight.detach().numpy().copy(),
            "rank/us": us, "rank/cands": cands, "rank/topk": np.int64(topk), "rank/preds": preds.astype(np.float32),
            "rank/full": full.astype(np.int64), "rank/predict": pred_pairs}


def ml100k_case(n_samples=12800):
    cwd = os.getcwd()
    os.chdir(G.REF)
    try:
        cfg = lg_config(num_ng=1, epochs=1, early_stop=False, algo_name="lightgcn", dataset="ml-100k")
        G.seed_all(cfg["seed"])
        df = RawDataReader(cfg).get_data()
        pre = Preprocessor(cfg)
        df = pre.process(df)
        cfg["user_num"], cfg["item_num"] = pre.user_num, pre.item_num
        tr_idx, te_idx = TestSplitter(cfg).split(df)
        train_set, test_set = df.iloc[tr_idx, :].copy(), df.iloc[te_idx, :].copy()
        test_ur, train_ur = get_ur(test_set), get_ur(train_set)
        cfg["train_ur"] = train_ur
        cfg["inter_matrix"] = get_inter_matrix(train_set, cfg)                 # test.py:88-89
        model = LightGCN(cfg)
        P0, Q0 = model.embed_user.weight.detach().numpy().copy(), model.embed_item.weight.detach().numpy().copy()
        samples = BasicNegtiveSampler(train_set, cfg).sampling()[:n_samples]
        loader = get_dataloader(BasicDataset(samples), batch_size=cfg["batch_size"], shuffle=True, num_workers=0)
        rng_state = torch.get_rng_state().numpy().copy()
        ref_abs.tqdm = G._TqdmCapture
        G._TqdmCapture.epoch_losses = []
        model.fit(loader)
        epoch_losses = np.array(G._TqdmCapture.epoch_losses, dtype=np.float64)
        test_u, test_ucands = build_candidates_set(test_ur, train_ur, cfg)
        cands = np.stack([c[1] for c in test_ucands]).astype(np.int64)
        preds = model.rank(get_dataloader(CandidatesDataset(test_ucands), batch_size=128, shuffle=False, num_workers=0))
    finally:
        os.chdir(cwd)
    print("ml-100k LightGCN: samples", samples.shape, "epoch losses", epoch_losses, "preds", preds.shape)
    return {"ml/meta": np.array([cfg["user_num"], cfg["item_num"], cfg["factors"], cfg["num_layers"]], dtype=np.int64),
            "ml/hyper": np.array([cfg["lr"], cfg["reg_1"], cfg["reg_2"]], dtype=np.float64),
            "ml/batch_size": np.int64(cfg["batch_size"]), "ml/topk": np.int64(cfg["topk"]), "ml/seed": np.int64(cfg["seed"]),
            "ml/train_users": train_set["user"].to_numpy().astype(np.int32),
            "ml/train_items": train_set["item"].to_numpy().astype(np.int32),
            "ml/samples": samples.astype(np.int32), "ml/rng_state_before_fit": rng_state, "ml/P0": P0, "ml/Q0": Q0,
            "ml/epoch_losses": epoch_losses, "ml/P1": model.embed_user.weight.detach().numpy().copy(),
            "ml/Q1": model.embed_item.weight.detach().numpy().copy(),
            "ml/test_u": np.array(test_u, dtype=np.int64), "ml/cands": cands, "ml/preds": preds.astype(np.float32)}


def main():
    rng = np.random.default_rng(2020)
    out, names = {}, []
    for (name, U, I, d, L, ne, B, lt, opt, reg, lr, ns) in [
        ("lg_bpr_adam", 50, 40, 64, 2, 600, 64, "BPR", "default", 0.0, 0.01, 3),      # lightgcn.yaml shape
        ("lg_bpr_l3_reg", 60, 50, 32, 3, 500, 96, "BPR", "default", 1e-3, 0.01, 3),
        ("lg_bpr_sgd", 50, 40, 16, 2, 300, 64, "BPR", "sgd", 1e-3, 0.05, 3),
        ("lg_tl_sgd_l1", 30, 40, 8, 1, 200, 48, "TL", "sgd", 1e-3, 0.05, 2),
        ("lg_cl_adam", 50, 40, 32, 2, 400, 64, "CL", "default", 1e-3, 0.01, 3),
    ]:
        out.update(kat_case(name, U, I, d, L, ne, B, lt, opt, reg, lr, ns, rng))
        names.append(name)
    out["names"] = np.array(names)
    out.update(adj_and_rank_case(rng))
    out.update(ml100k_case())
    np.savez_compressed(os.path.join(HERE, "kat_lightgcn.npz"), **out)
    print("kat_lightgcn.npz:", names)


if __name__ == "__main__":
    main()
