#!/usr/bin/env python
"""Golden vectors for the FM widening, generated from the REAL reference
(`daisy.model.FMRecommender.FM`, imported from /root/reference; nothing is copied).  Runs only in
the build container; the output tests/golden/kat_fm.npz is committed.

    python tests/golden/make_golden_fm.py

Contents:
  (1) step KATs: random tables AND non-zero biases, batches with duplicate users / items, through
      FM.calc_loss -> backward -> optimizer.step for BPR / TL (the global bias only matters there
      and in the point-wise losses) / CL / SL with SGD, and BPR with dense Adam
      (FMRecommender.py:61-93, AbstractRecommender.py:48-67,119-126);
  (2) FM.rank / full_rank / predict on random tables with biases (FMRecommender.py:95-133);
  (3) ml-100k end to end in run_examples/test.py's call order with --algo_name fm (fm.yaml:
      factors 84, lr 0.001, SGD) for 2 epochs: epoch losses, final parameters, ranked lists.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G  # noqa: E402  (sets up the shims and the reference import path)

import torch  # noqa: E402
import yaml  # noqa: E402
from daisy.model.FMRecommender import FM  # noqa: E402
import daisy.model.AbstractRecommender as ref_abs  # noqa: E402
from daisy.utils.dataset import BasicDataset, CandidatesDataset, get_dataloader  # noqa: E402
from daisy.utils.loader import Preprocessor, RawDataReader  # noqa: E402
from daisy.utils.sampler import BasicNegtiveSampler  # noqa: E402
from daisy.utils.splitter import TestSplitter  # noqa: E402
from daisy.utils.utils import build_candidates_set, get_ur  # noqa: E402


def fm_config(**over):
    cfg = G.base_config()
    cfg.update(yaml.safe_load(open(os.path.join(G.REF, "daisy/assets/fm.yaml"))))
    cfg.update(over)
    return cfg


def _params(model):
    return [model.embed_user.weight, model.embed_item.weight, model.u_bias.weight, model.i_bias.weight,
            model.bias_]


def kat_case(name, U, I, d, B, loss_type, optimizer, reg, lr, n_steps, rng, scale=0.3):
    cfg = fm_config(user_num=U, item_num=I, factors=d, loss_type=loss_type, optimizer=optimizer,
                    reg_1=reg, reg_2=reg, lr=lr, epochs=1, early_stop=False, init_method="default")
    model = FM(cfg)
    init = [(rng.standard_normal((U, d)) * scale).astype(np.float32),
            (rng.standard_normal((I, d)) * scale).astype(np.float32),
            (rng.standard_normal((U, 1)) * 0.2).astype(np.float32),
            (rng.standard_normal((I, 1)) * 0.2).astype(np.float32),
            np.array([0.15], dtype=np.float32)]
    with torch.no_grad():
        for p, v in zip(_params(model), init):
            p.copy_(torch.from_numpy(v))
    opt = model._build_optimizer(optimizer=model.optimizer, lr=model.lr)
    model.criterion = model._build_criterion(model.loss_type)
    us, is_, js, losses = [], [], [], []
    hist = [[] for _ in range(5)]
    for _ in range(n_steps):
        u = rng.integers(0, U, size=B).astype(np.int32)
        i = rng.integers(0, I, size=B).astype(np.int32)
        j = (rng.integers(0, 2, size=B) if loss_type in ("CL", "SL") else rng.integers(0, I, size=B)).astype(np.int32)
        u[1] = u[0]
        i[2] = i[0]
        if loss_type not in ("CL", "SL"):
            j[3] = i[0]
        model.zero_grad()
        loss = model.calc_loss([torch.from_numpy(x) for x in (u, i, j)])
        loss.backward()
        opt.step()
        us.append(u); is_.append(i); js.append(j)
        losses.append(float(loss.item()))
        for h, p in zip(hist, _params(model)):
            h.append(p.detach().numpy().copy())
    out = {f"{name}/meta": np.array([U, I, d, B, n_steps], dtype=np.int64),
           f"{name}/hyper": np.array([lr, reg, reg], dtype=np.float64),
           f"{name}/loss_type": np.array(loss_type), f"{name}/optimizer": np.array(optimizer),
           f"{name}/u": np.stack(us), f"{name}/i": np.stack(is_), f"{name}/j": np.stack(js),
           f"{name}/loss": np.array(losses, dtype=np.float64)}
    for k, key in enumerate(("P", "Q", "bu", "bi", "b0")):
        out[f"{name}/{key}0"] = init[k]
        out[f"{name}/{key}"] = np.stack(hist[k])
    return out


def rank_case(rng):
    U, I, d, C, nB, topk = 40, 300, 20, 100, 10, 10
    model = FM(fm_config(user_num=U, item_num=I, factors=d, topk=topk))
    init = [(rng.standard_normal((U, d)) * 0.1).astype(np.float32),
            (rng.standard_normal((I, d)) * 0.1).astype(np.float32),
            (rng.standard_normal((U, 1)) * 0.05).astype(np.float32),
            (rng.standard_normal((I, 1)) * 0.05).astype(np.float32),
            np.array([-0.2], dtype=np.float32)]
    with torch.no_grad():
        for p, v in zip(_params(model), init):
            p.copy_(torch.from_numpy(v))
    us = rng.integers(0, U, size=nB).astype(np.int64)
    cands = rng.integers(0, I, size=(nB, C)).astype(np.int64)
    loader = get_dataloader(CandidatesDataset([[int(us[b]), cands[b]] for b in range(nB)]), batch_size=4,
                            shuffle=False, num_workers=0)
    preds = model.rank(loader)
    full = np.stack([model.full_rank(int(u)) for u in us])
    pred_pairs = np.array([model.predict(int(us[b]), int(cands[b, 0])) for b in range(nB)], dtype=np.float32)
    out = {"rank/us": us, "rank/cands": cands, "rank/topk": np.int64(topk),
           "rank/preds": preds.astype(np.float32), "rank/full": full.astype(np.int64),
           "rank/predict": pred_pairs}
    for k, key in enumerate(("P", "Q", "bu", "bi", "b0")):
        out[f"rank/{key}"] = init[k]
    return out


def ml100k_case(epochs=2):
    cwd = os.getcwd()
    os.chdir(G.REF)
    try:
        cfg = fm_config(num_ng=1, epochs=epochs, early_stop=False, algo_name="fm", dataset="ml-100k")
        G.seed_all(cfg["seed"])
        df = RawDataReader(cfg).get_data()
        pre = Preprocessor(cfg)
        df = pre.process(df)
        cfg["user_num"], cfg["item_num"] = pre.user_num, pre.item_num
        tr_idx, te_idx = TestSplitter(cfg).split(df)
        train_set, test_set = df.iloc[tr_idx, :].copy(), df.iloc[te_idx, :].copy()
        test_ur, train_ur = get_ur(test_set), get_ur(train_set)
        cfg["train_ur"] = train_ur
        model = FM(cfg)
        init = [p.detach().numpy().copy() for p in _params(model)]
        samples = BasicNegtiveSampler(train_set, cfg).sampling()
        loader = get_dataloader(BasicDataset(samples), batch_size=cfg["batch_size"], shuffle=True, num_workers=0)
        rng_state = torch.get_rng_state().numpy().copy()
        ref_abs.tqdm = G._TqdmCapture
        G._TqdmCapture.epoch_losses = []
        model.fit(loader)
        epoch_losses = np.array(G._TqdmCapture.epoch_losses, dtype=np.float64)
        final = [p.detach().numpy().copy() for p in _params(model)]
        test_u, test_ucands = build_candidates_set(test_ur, train_ur, cfg)
        cands = np.stack([c[1] for c in test_ucands]).astype(np.int64)
        preds = model.rank(get_dataloader(CandidatesDataset(test_ucands), batch_size=128, shuffle=False,
                                          num_workers=0))
    finally:
        os.chdir(cwd)
    out = {"ml/user_num": np.int64(cfg["user_num"]), "ml/item_num": np.int64(cfg["item_num"]),
           "ml/hyper": np.array([cfg["lr"], cfg["reg_1"], cfg["reg_2"]], dtype=np.float64),
           "ml/factors": np.int64(cfg["factors"]), "ml/batch_size": np.int64(cfg["batch_size"]),
           "ml/epochs": np.int64(epochs), "ml/topk": np.int64(cfg["topk"]), "ml/seed": np.int64(cfg["seed"]),
           "ml/samples": samples.astype(np.int32), "ml/rng_state_before_fit": rng_state,
           "ml/epoch_losses": epoch_losses, "ml/test_u": np.array(test_u, dtype=np.int64), "ml/cands": cands,
           "ml/preds": preds.astype(np.float32)}
    for k, key in enumerate(("P", "Q", "bu", "bi", "b0")):
        out[f"ml/{key}0"] = init[k]
        out[f"ml/{key}1"] = final[k]
    print("ml-100k FM: samples", samples.shape, "epoch losses", epoch_losses, "preds", preds.shape)
    return out


def main():
    rng = np.random.default_rng(84)
    out, names = {}, []
    for (name, U, I, d, B, lt, opt, reg, lr, ns) in [
        ("fm_bpr_d32", 50, 40, 32, 64, "BPR", "sgd", 1e-3, 0.01, 3),
        ("fm_bpr_d84", 120, 90, 84, 200, "BPR", "sgd", 1e-3, 0.01, 2),      # fm.yaml's factor count
        ("fm_tl_d32", 50, 40, 32, 64, "TL", "sgd", 1e-3, 0.01, 2),
        ("fm_hl_d16", 30, 40, 16, 48, "HL", "sgd", 1e-3, 0.01, 2),
        ("fm_cl_d32", 50, 40, 32, 64, "CL", "sgd", 1e-3, 0.01, 3),
        ("fm_sl_d64", 50, 40, 64, 96, "SL", "sgd", 1e-3, 0.01, 2),
        ("fm_bpr_adam", 50, 40, 32, 64, "BPR", "adam", 1e-3, 0.01, 4),
        ("fm_cl_adam", 50, 40, 32, 64, "CL", "adam", 0.0, 0.01, 3),
    ]:
        out.update(kat_case(name, U, I, d, B, lt, opt, reg, lr, ns, rng))
        names.append(name)
    out["names"] = np.array(names)
    out.update(rank_case(rng))
    out.update(ml100k_case())
    np.savez_compressed(os.path.join(HERE, "kat_fm.npz"), **out)
    print("kat_fm.npz:", names)


if __name__ == "__main__":
    main()
