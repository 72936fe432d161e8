#!/usr/bin/env python
"""Golden vectors for the NeuMF widening, generated from the REAL reference
(`daisy.model.NeuMFRecommender.NeuMF`, imported from /root/reference; nothing is copied).  Runs
only in the build container; the output tests/golden/kat_neumf.npz is committed.

    python tests/golden/make_golden_neumf.py

All cases use dropout = 0: the reference draws dropout masks from torch's global generator, which
a device generator cannot replay (see oracle/neumf_numpy.py).
  (1) step KATs through NeuMF.calc_loss -> backward -> optimizer.step (NeuMFRecommender.py:118-169,
      AbstractRecommender.py:48-67,119-126): model NeuMF / GMF / MLP, BPR / TL / CL, Adam (the
      model's default) and SGD, 2 and 3 MLP layers, batches with duplicate users and items;
  (2) NeuMF.rank / full_rank / predict on random parameters (:171-233);
  (3) ml-100k end to end in run_examples/test.py's call order with --algo_name neumf
      (neumf.yaml: factors 24, num_layers 2, lr 0.001, Adam; and with --optimizer sgd), dropout 0,
      one epoch over the first 25 600 triples (100 batches; see ml100k_case for why not more).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G  # noqa: E402

import torch  # noqa: E402
import yaml  # noqa: E402
from daisy.model.NeuMFRecommender import NeuMF  # noqa: E402
import daisy.model.AbstractRecommender as ref_abs  # noqa: E402
from daisy.utils.dataset import BasicDataset, CandidatesDataset, get_dataloader  # noqa: E402
from daisy.utils.loader import Preprocessor, RawDataReader  # noqa: E402
from daisy.utils.sampler import BasicNegtiveSampler  # noqa: E402
from daisy.utils.splitter import TestSplitter  # noqa: E402
from daisy.utils.utils import build_candidates_set, get_ur  # noqa: E402


def neumf_config(**over):
    cfg = G.base_config()
    cfg.update(yaml.safe_load(open(os.path.join(G.REF, "daisy/assets/neumf.yaml"))))
    cfg["dropout"] = 0.0
    cfg.update(over)
    return cfg


def named_params(model):
    """Parameters under the oracle's names (oracle/neumf_numpy.py: param_names)."""
    out = {"uG": model.embed_user_GMF.weight, "iG": model.embed_item_GMF.weight,
           "uM": model.embed_user_MLP.weight, "iM": model.embed_item_MLP.weight}
    lin = [m for m in model.MLP_layers if isinstance(m, torch.nn.Linear)]
    for l, m in enumerate(lin, 1):
        out[f"W{l}"], out[f"b{l}"] = m.weight, m.bias
    out["Wp"], out["bp"] = model.predict_layer.weight, model.predict_layer.bias
    return out


def kat_case(name, U, I, d, L, B, model_name, loss_type, optimizer, reg, lr, n_steps, rng):
    cfg = neumf_config(user_num=U, item_num=I, factors=d, num_layers=L, model_name=model_name,
                       loss_type=loss_type, optimizer=optimizer, reg_1=reg, reg_2=reg, lr=lr,
                       epochs=1, early_stop=False, init_method="default")
    torch.manual_seed(int(rng.integers(1 << 30)))
    model = NeuMF(cfg)
    model.train()
    with torch.no_grad():          # make every bias / both signs of the ReLU matter
        for k, p in named_params(model).items():
            if k.startswith("b"):
                p.copy_(torch.from_numpy((rng.standard_normal(p.shape) * 0.05).astype(np.float32)))
    init = {k: p.detach().numpy().copy() for k, p in named_params(model).items()}
    opt = model._build_optimizer(optimizer=model.optimizer, lr=model.lr)
    model.criterion = model._build_criterion(model.loss_type)
    us, is_, js, losses = [], [], [], []
    hist = {k: [] for k in init}
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
        for k, p in named_params(model).items():
            hist[k].append(p.detach().numpy().copy())
    out = {f"{name}/meta": np.array([U, I, d, L, B, n_steps], dtype=np.int64),
           f"{name}/hyper": np.array([lr, reg, reg], dtype=np.float64),
           f"{name}/model": np.array(model_name), f"{name}/loss_type": np.array(loss_type),
           f"{name}/optimizer": np.array(model.optimizer),
           f"{name}/u": np.stack(us), f"{name}/i": np.stack(is_), f"{name}/j": np.stack(js),
           f"{name}/loss": np.array(losses, dtype=np.float64)}
    for k in init:
        out[f"{name}/{k}0"] = init[k]
        out[f"{name}/{k}"] = hist[k][-1]              # parameters after the last step
    return out


def rank_case(rng):
    U, I, d, L, C, nB, topk = 40, 300, 16, 3, 100, 10, 10
    torch.manual_seed(5)
    model = NeuMF(neumf_config(user_num=U, item_num=I, factors=d, num_layers=L, topk=topk))
    with torch.no_grad():
        for k, p in named_params(model).items():
            if k.startswith("b"):
                p.copy_(torch.from_numpy((rng.standard_normal(p.shape) * 0.05).astype(np.float32)))
    model.eval()
    us = rng.integers(0, U, size=nB).astype(np.int64)
    cands = rng.integers(0, I, size=(nB, C)).astype(np.int64)
    loader = get_dataloader(CandidatesDataset([[int(us[b]), cands[b]] for b in range(nB)]), batch_size=4,
                            shuffle=False, num_workers=0)
    preds = model.rank(loader)
    full = np.stack([model.full_rank(int(u)) for u in us])
    pred_pairs = np.array([model.predict(int(us[b]), int(cands[b, 0])) for b in range(nB)], dtype=np.float32)
    out = {"rank/meta": np.array([U, I, d, L], dtype=np.int64), "rank/us": us, "rank/cands": cands,
           "rank/topk": np.int64(topk), "rank/preds": preds.astype(np.float32),
           "rank/full": full.astype(np.int64), "rank/predict": pred_pairs}
    for k, p in named_params(model).items():
        out[f"rank/{k}"] = p.detach().numpy().copy()
    return out


def ml100k_case(epochs=2, prefix="ml", **over):
    cwd = os.getcwd()
    os.chdir(G.REF)
    try:
        cfg = neumf_config(num_ng=1, epochs=epochs, early_stop=False, algo_name="neumf", dataset="ml-100k", **over)
        G.seed_all(cfg["seed"])
        df = RawDataReader(cfg).get_data()
        pre = Preprocessor(cfg)
        df = pre.process(df)
        cfg["user_num"], cfg["item_num"] = pre.user_num, pre.item_num
        tr_idx, te_idx = TestSplitter(cfg).split(df)
        train_set, test_set = df.iloc[tr_idx, :].copy(), df.iloc[te_idx, :].copy()
        test_ur, train_ur = get_ur(test_set), get_ur(train_set)
        cfg["train_ur"] = train_ur
        model = NeuMF(cfg)
        init = {k: p.detach().numpy().copy() for k, p in named_params(model).items()}
        # 100 batches: NeuMF's training dynamics amplify last-ulp differences (fp32 summation order)
        # by orders of magnitude over a full 307-step epoch - two runs of ANY fp32 implementation with
        # different reduction orders end on different branches - so the end-to-end vectors stop where
        # round-off is still round-off (measured: 1e-8 at step 100, >1e-5 after ~170 steps)
        samples = BasicNegtiveSampler(train_set, cfg).sampling()[:25600]
        loader = get_dataloader(BasicDataset(samples), batch_size=cfg["batch_size"], shuffle=True, num_workers=0)
        rng_state = torch.get_rng_state().numpy().copy()
        ref_abs.tqdm = G._TqdmCapture
        G._TqdmCapture.epoch_losses = []
        model.fit(loader)
        epoch_losses = np.array(G._TqdmCapture.epoch_losses, dtype=np.float64)
        final = {k: p.detach().numpy().copy() for k, p in named_params(model).items()}
        test_u, test_ucands = build_candidates_set(test_ur, train_ur, cfg)
        cands = np.stack([c[1] for c in test_ucands]).astype(np.int64)
        preds = model.rank(get_dataloader(CandidatesDataset(test_ucands), batch_size=128, shuffle=False,
                                          num_workers=0))
    finally:
        os.chdir(cwd)
    out = {"meta": np.array([cfg["user_num"], cfg["item_num"], cfg["factors"], cfg["num_layers"]], dtype=np.int64),
           "hyper": np.array([cfg["lr"], cfg["reg_1"], cfg["reg_2"]], dtype=np.float64),
           "optimizer": np.array(model.optimizer),
           "batch_size": np.int64(cfg["batch_size"]), "epochs": np.int64(epochs),
           "topk": np.int64(cfg["topk"]), "seed": np.int64(cfg["seed"]),
           "samples": samples.astype(np.int32), "rng_state_before_fit": rng_state,
           "epoch_losses": epoch_losses, "test_u": np.array(test_u, dtype=np.int64), "cands": cands,
           "preds": preds.astype(np.float32)}
    for k in init:
        out[f"{k}0"] = init[k]
        out[f"{k}1"] = final[k]
    print(f"ml-100k NeuMF ({model.optimizer}): samples", samples.shape, "epoch losses", epoch_losses, "preds",
          preds.shape)
    return {f"{prefix}/{k}": v for k, v in out.items()}


def main():
    rng = np.random.default_rng(2017)
    out, names = {}, []
    for (name, U, I, d, L, B, model, lt, opt, reg, lr, ns) in [
        ("neumf_bpr_adam", 50, 40, 24, 2, 64, "NeuMF", "BPR", "default", 1e-3, 0.001, 3),   # neumf.yaml shape
        ("neumf_bpr_l3", 60, 50, 16, 3, 96, "NeuMF", "BPR", "default", 1e-3, 0.001, 2),
        ("neumf_bpr_sgd", 50, 40, 8, 2, 64, "NeuMF", "BPR", "sgd", 1e-3, 0.05, 3),
        ("neumf_tl_sgd", 50, 40, 8, 2, 64, "NeuMF", "TL", "sgd", 1e-3, 0.05, 2),
        ("neumf_cl_adam", 50, 40, 24, 2, 64, "NeuMF", "CL", "default", 1e-3, 0.001, 3),
        ("gmf_bpr_adam", 50, 40, 24, 2, 64, "GMF", "BPR", "default", 1e-3, 0.001, 2),
        ("mlp_bpr_adam", 50, 40, 24, 2, 64, "MLP", "BPR", "default", 1e-3, 0.001, 2),
        ("neumf_noreg_d64", 80, 70, 64, 3, 128, "NeuMF", "BPR", "default", 0.0, 0.001, 2),   # BASELINE config 4 shape
    ]:
        out.update(kat_case(name, U, I, d, L, B, model, lt, opt, reg, lr, ns, rng))
        names.append(name)
    out["names"] = np.array(names)
    out.update(rank_case(rng))
    out.update(ml100k_case(epochs=1))                                          # neumf.yaml: Adam, lr 0.001
    sgd = ml100k_case(epochs=1, prefix="mlsgd", optimizer="sgd", lr=0.001)       # smooth optimiser at a stable step size
    for k in ("samples", "cands", "test_u"):                                     # same data as ml/*
        assert np.array_equal(sgd[f"mlsgd/{k}"], out[f"ml/{k}"])
        del sgd[f"mlsgd/{k}"]
    out.update(sgd)
    np.savez_compressed(os.path.join(HERE, "kat_neumf.npz"), **out)
    print("kat_neumf.npz:", names)


if __name__ == "__main__":
    main()
