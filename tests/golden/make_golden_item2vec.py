#!/usr/bin/env python
"""Golden vectors for the Item2Vec widening (rest of SURVEY.md §8f rank 4), generated from the REAL
reference (`daisy.model.Item2VecRecommender.Item2Vec` + `SkipGramNegativeSampler`, imported from
/root/reference; nothing is copied).  Runs only in the build container.

    python tests/golden/make_golden_item2vec.py        # -> tests/golden/kat_item2vec.npz

  (1) step KATs through Item2Vec.calc_loss -> backward -> optimizer.step (Item2VecRecommender.py:47-69):
      Adam (the model's default) and SGD, batches where an item is target and context at once;
  (2) ml-100k in run_examples/test.py's call order with --algo_name item2vec (item2vec.yaml: d=100,
      lr 0.001, Adam, context_window 2): SkipGramNegativeSampler triples (first 12 800 = 50 batches),
      fit incl. the user-embedding build (:53-59), rank / full_rank / predict.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G  # noqa: E402

import pandas as pd  # noqa: E402
import torch  # noqa: E402
import yaml  # noqa: E402

if not hasattr(pd.Series, "iteritems"):          # sampler.py:136 (pandas >= 2 renamed it)
    pd.Series.iteritems = pd.Series.items

from daisy.model.Item2VecRecommender import Item2Vec  # noqa: E402
import daisy.model.AbstractRecommender as ref_abs  # noqa: E402
from daisy.utils.dataset import BasicDataset, CandidatesDataset, get_dataloader  # noqa: E402
from daisy.utils.loader import Preprocessor, RawDataReader  # noqa: E402
from daisy.utils.sampler import SkipGramNegativeSampler  # noqa: E402
from daisy.utils.splitter import TestSplitter  # noqa: E402
from daisy.utils.utils import build_candidates_set, get_ur  # noqa: E402


def i2v_config(**over):
    cfg = G.base_config()
    cfg.update(yaml.safe_load(open(os.path.join(G.REF, "daisy/assets/item2vec.yaml"))))
    cfg.update(over)
    return cfg


def kat_case(name, U, I, d, B, optimizer, lr, n_steps, rng):
    cfg = i2v_config(user_num=U, item_num=I, factors=d, optimizer=optimizer, lr=lr, epochs=1, early_stop=False,
                     init_method="default", train_ur={})
    torch.manual_seed(int(rng.integers(1 << 30)))
    model = Item2Vec(cfg)
    with torch.no_grad():
        model.shared_embedding.weight.mul_(30.0)            # normal(0, 0.01) init: make the logits matter
    S0 = model.shared_embedding.weight.detach().numpy().copy()
    opt = model._build_optimizer(optimizer=model.optimizer, lr=model.lr)
    model.criterion = model._build_criterion(model.loss_type)
    ts, cs, ys, losses = [], [], [], []
    for _ in range(n_steps):
        t = rng.integers(0, I, size=B).astype(np.int32)
        c = rng.integers(0, I, size=B).astype(np.int32)
        y = rng.integers(0, 2, size=B).astype(np.int32)
        t[1] = t[0]; c[2] = c[0]; c[3] = t[0]; t[4] = c[5]; c[6] = t[6]      # duplicates, target==context rows
        model.zero_grad()
        loss = model.calc_loss([torch.from_numpy(x) for x in (t, c, y)])
        loss.backward()
        opt.step()
        ts.append(t); cs.append(c); ys.append(y)
        losses.append(float(loss.item()))
    return {f"{name}/meta": np.array([U, I, d, B, n_steps], dtype=np.int64), f"{name}/lr": np.float64(lr),
            f"{name}/optimizer": np.array(model.optimizer), f"{name}/S0": S0,
            f"{name}/t": np.stack(ts), f"{name}/c": np.stack(cs), f"{name}/y": np.stack(ys),
            f"{name}/loss": np.array(losses, dtype=np.float64),
            f"{name}/S": model.shared_embedding.weight.detach().numpy().copy()}


def ml100k_case(n_samples=12800):
    cwd = os.getcwd()
    os.chdir(G.REF)
    try:
        cfg = i2v_config(epochs=1, early_stop=False, algo_name="item2vec", dataset="ml-100k")
        G.seed_all(cfg["seed"])
        df = RawDataReader(cfg).get_data()
        pre = Preprocessor(cfg)
        df = pre.process(df)
        cfg["user_num"], cfg["item_num"] = pre.user_num, pre.item_num
        tr_idx, te_idx = TestSplitter(cfg).split(df)
        train_set, test_set = df.iloc[tr_idx, :].copy(), df.iloc[te_idx, :].copy()
        test_ur, train_ur = get_ur(test_set), get_ur(train_set)
        cfg["train_ur"] = train_ur
        model = Item2Vec(cfg)                                                       # test.py:98
        S0 = model.shared_embedding.weight.detach().numpy().copy()
        Uemb0 = model.user_embedding.weight.detach().numpy().copy()
        samples = SkipGramNegativeSampler(train_set, cfg).sampling()                # test.py:99-100
        rng = np.random.default_rng(0)
        samples = samples[rng.permutation(len(samples))[:n_samples]].astype(np.int32)   # a random subset (100 batches)
        loader = get_dataloader(BasicDataset(samples), batch_size=cfg["batch_size"], shuffle=True, num_workers=0)
        rng_state = torch.get_rng_state().numpy().copy()
        ref_abs.tqdm = G._TqdmCapture
        G._TqdmCapture.epoch_losses = []
        model.fit(loader)
        epoch_losses = np.array(G._TqdmCapture.epoch_losses, dtype=np.float64)
        test_u, test_ucands = build_candidates_set(test_ur, train_ur, cfg)
        cands = np.stack([c[1] for c in test_ucands]).astype(np.int64)
        preds = model.rank(get_dataloader(CandidatesDataset(test_ucands), batch_size=128, shuffle=False, num_workers=0))
        full = np.stack([model.full_rank(int(u)) for u in test_u[:8]])
        pp = np.array([model.predict(int(u), 5) for u in test_u[:8]], dtype=np.float32)
    finally:
        os.chdir(cwd)
    ur_users = np.concatenate([[u] * len(v) for u, v in train_ur.items()]).astype(np.int32)
    ur_items = np.concatenate([sorted(v) for v in train_ur.values()]).astype(np.int32)
    print("ml-100k Item2Vec: samples", samples.shape, "epoch losses", epoch_losses, "preds", preds.shape)
    return {"ml/meta": np.array([cfg["user_num"], cfg["item_num"], cfg["factors"]], dtype=np.int64),
            "ml/lr": np.float64(cfg["lr"]), "ml/batch_size": np.int64(cfg["batch_size"]), "ml/topk": np.int64(cfg["topk"]),
            "ml/seed": np.int64(cfg["seed"]), "ml/samples": samples, "ml/rng_state_before_fit": rng_state,
            "ml/S0": S0, "ml/Uemb0": Uemb0, "ml/ur_users": ur_users, "ml/ur_items": ur_items,
            "ml/epoch_losses": epoch_losses, "ml/S1": model.shared_embedding.weight.detach().numpy().copy(),
            "ml/Uemb1": model.user_embedding.weight.detach().numpy().copy(),
            "ml/test_u": np.array(test_u, dtype=np.int64), "ml/cands": cands, "ml/preds": preds.astype(np.float32),
            "ml/full8": full.astype(np.int64), "ml/predict8": pp}


def main():
    rng = np.random.default_rng(2016)
    out, names = {}, []
    for (name, U, I, d, B, opt, lr, ns) in [
        ("i2v_adam_d100", 20, 60, 100, 64, "default", 0.001, 3),       # item2vec.yaml shape
        ("i2v_sgd_d32", 20, 50, 32, 96, "sgd", 0.05, 3),
        ("i2v_adam_d8_b1", 5, 9, 8, 8, "default", 0.01, 2),
    ]:
        out.update(kat_case(name, U, I, d, B, opt, lr, ns, rng))
        names.append(name)
    out["names"] = np.array(names)
    out.update(ml100k_case())
    np.savez_compressed(os.path.join(HERE, "kat_item2vec.npz"), **out)
    print("kat_item2vec.npz:", names)


if __name__ == "__main__":
    main()
