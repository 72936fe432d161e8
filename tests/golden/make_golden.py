#!/usr/bin/env python
"""Generate the golden vectors that pin `oracle/` (and through it the HIP path)
to the REAL reference.  Runs ONLY in the build container, where the reference
checkout is mounted read-only at /root/reference; the outputs (small .npz files
in this directory) are committed and travel to the GPU box, which has no
reference checkout.

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz

What is imported from the reference (nothing is copied):
    daisy.model.MFRecommender.MF, daisy.utils.{loader,splitter,sampler,dataset,utils}
and what is run through them:
  (1) kat_steps.npz   — per-step known answers: random tables + batches (with
      duplicate rows, an item that is positive in one sample and negative in
      another, all-zero rows, reg=0, HL/TL losses, SGD and dense Adam) through
      MF.calc_loss -> backward -> optimizer.step           (MFRecommender.py:70-97,
      AbstractRecommender.py:48-67,119-126)
  (2) ml100k_c1.npz   — BASELINE config C1 end to end in run_examples/test.py's
      call order (test.py:43-120): loader -> 10filter -> tsbr split -> get_ur ->
      MF(config) -> BasicNegtiveSampler.sampling -> BasicDataset/get_dataloader ->
      MF.fit (epoch losses captured from the tqdm postfix) -> build_candidates_set
      -> MF.rank.
  (3) rank_kat.npz    — MF.rank / MF.full_rank on random tables incl. duplicate
      candidates (MFRecommender.py:106-133).
"""
import logging
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("DAISY_REFERENCE", "/root/reference")
sys.path.insert(0, os.path.join(HERE, "_shims"))
sys.path.insert(0, REF)
if not hasattr(np, "asfarray"):          # daisy/utils/metrics.py:206 (numpy>=2 removed it)
    np.asfarray = lambda a, dtype=np.float64: np.asarray(a, dtype=dtype)

import torch  # noqa: E402
import yaml  # noqa: E402

torch.set_num_threads(1)                  # deterministic CPU reductions

from daisy.model.MFRecommender import MF  # noqa: E402
import daisy.model.AbstractRecommender as ref_abs  # noqa: E402
from daisy.utils.dataset import BasicDataset, CandidatesDataset, get_dataloader  # noqa: E402
from daisy.utils.loader import Preprocessor, RawDataReader  # noqa: E402
from daisy.utils.sampler import BasicNegtiveSampler  # noqa: E402
from daisy.utils.splitter import TestSplitter  # noqa: E402
from daisy.utils.utils import build_candidates_set, get_ur  # noqa: E402


def base_config(**over):
    cfg = {}
    cfg.update(yaml.safe_load(open(os.path.join(REF, "daisy/assets/basic.yaml"))))
    cfg.update(yaml.safe_load(open(os.path.join(REF, "daisy/assets/mf.yaml"))))
    cfg["logger"] = logging.getLogger("golden")
    cfg.update(over)
    return cfg


def seed_all(seed):
    import random
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)


# ----------------------------------------------------------------------------
def kat_case(name, U, I, d, B, loss_type, optimizer, reg_1, reg_2, lr, n_steps, rng,
             zero_rows=False, scale=0.1):
    cfg = base_config(user_num=U, item_num=I, factors=d, loss_type=loss_type,
                      optimizer=optimizer, reg_1=reg_1, reg_2=reg_2, lr=lr,
                      epochs=1, early_stop=False, init_method="default")
    model = MF(cfg)
    P0 = (rng.standard_normal((U, d)) * scale).astype(np.float32)
    Q0 = (rng.standard_normal((I, d)) * scale).astype(np.float32)
    if zero_rows:
        P0[0] = 0.0
        Q0[:2] = 0.0
    with torch.no_grad():
        model.embed_user.weight.copy_(torch.from_numpy(P0))
        model.embed_item.weight.copy_(torch.from_numpy(Q0))
    opt = model._build_optimizer(optimizer=model.optimizer, lr=model.lr)
    model.criterion = model._build_criterion(model.loss_type)
    us, is_, js, losses, Ps, Qs = [], [], [], [], [], []
    for s in range(n_steps):
        u = rng.integers(0, U, size=B).astype(np.int32)
        i = rng.integers(0, I, size=B).astype(np.int32)
        j = rng.integers(0, I, size=B).astype(np.int32)
        if loss_type in ("CL", "SL"):    # point-wise rows are (user, item, label)  (sampler.py:93-98)
            j = rng.integers(0, 2, size=B).astype(np.int32)
        if B >= 4:                       # force the hard cases
            u[1] = u[0]                  # duplicate user
            i[2] = i[0]                  # duplicate positive
            if loss_type not in ("CL", "SL"):
                j[3] = i[0]              # same item positive and negative in one batch
        if zero_rows:
            u[0] = 0
            i[0] = 0
            j[0] = 1
        batch = [torch.from_numpy(x) for x in (u, i, j)]
        model.zero_grad()
        loss = model.calc_loss(batch)
        loss.backward()
        opt.step()
        us.append(u); is_.append(i); js.append(j)
        losses.append(float(loss.item()))
        Ps.append(model.embed_user.weight.detach().numpy().copy())
        Qs.append(model.embed_item.weight.detach().numpy().copy())
    return {
        f"{name}/meta": np.array([U, I, d, B, n_steps], dtype=np.int64),
        f"{name}/hyper": np.array([lr, reg_1, reg_2], dtype=np.float64),
        f"{name}/loss_type": np.array(loss_type), f"{name}/optimizer": np.array(optimizer),
        f"{name}/P0": P0, f"{name}/Q0": Q0,
        f"{name}/u": np.stack(us), f"{name}/i": np.stack(is_), f"{name}/j": np.stack(js),
        f"{name}/loss": np.array(losses, dtype=np.float64),
        f"{name}/P": np.stack(Ps), f"{name}/Q": np.stack(Qs),
    }


def make_kat_steps():
    rng = np.random.default_rng(20220925)
    out, names = {}, []
    cases = [
        # name          U    I    d   B    loss  opt     reg1   reg2   lr    steps
        ("bpr_d32",     50,  40,  32, 64,  "BPR", "sgd",  1e-3,  1e-3,  0.01, 3, {}),
        ("bpr_d64",     300, 200, 64, 256, "BPR", "sgd",  1e-3,  1e-3,  0.01, 3, {}),
        ("bpr_d100",    60,  70,  100, 33, "BPR", "sgd",  1e-3,  1e-3,  0.01, 2, {}),
        ("bpr_d8",      20,  30,  8,  17,  "BPR", "sgd",  1e-2,  1e-2,  0.05, 2, {}),
        ("bpr_noreg",   50,  40,  32, 64,  "BPR", "sgd",  0.0,   0.0,   0.01, 2, {}),
        ("bpr_zero",    50,  40,  32, 64,  "BPR", "sgd",  1e-3,  1e-3,  0.01, 2, {"zero_rows": True}),
        ("bpr_b1",      10,  10,  16, 1,   "BPR", "sgd",  1e-3,  1e-3,  0.01, 2, {}),
        ("bpr_big",     50,  40,  32, 64,  "BPR", "sgd",  1e-3,  1e-3,  0.01, 2, {"scale": 3.0}),
        ("hl_d32",      50,  40,  32, 64,  "HL",  "sgd",  1e-3,  1e-3,  0.01, 2, {"scale": 0.5}),
        ("tl_d32",      50,  40,  32, 64,  "TL",  "sgd",  1e-3,  1e-3,  0.01, 2, {"scale": 0.5}),
        ("bpr_adam",    50,  40,  32, 64,  "BPR", "adam", 1e-3,  1e-3,  0.01, 4, {}),
        ("cl_d32",      50,  40,  32, 64,  "CL",  "sgd",  1e-3,  1e-3,  0.01, 3, {"scale": 0.5}),
        ("sl_d64",      50,  40,  64, 96,  "SL",  "sgd",  1e-3,  1e-3,  0.01, 3, {"scale": 0.5}),
        ("cl_zero",     50,  40,  32, 64,  "CL",  "sgd",  1e-3,  1e-3,  0.01, 2, {"zero_rows": True}),
    ]
    for (name, U, I, d, B, lt, opt, r1, r2, lr, ns, kw) in cases:
        out.update(kat_case(name, U, I, d, B, lt, opt, r1, r2, lr, ns, rng, **kw))
        names.append(name)
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "kat_steps.npz"), **out)
    print("kat_steps.npz:", names)


def make_kat_optimizers():
    """kat_optimizers.npz: the remaining branches of _build_optimizer (AbstractRecommender.py:56-61) -
    optim.Adagrad / optim.RMSprop with torch's defaults, and what optim.SparseAdam does on the reference's
    dense nn.Embedding gradients (it refuses them)."""
    rng = np.random.default_rng(20220926)
    out, names = {}, []
    for (name, opt, lr, ns) in (("bpr_adagrad", "adagrad", 0.01, 4), ("bpr_rmsprop", "rmsprop", 0.001, 4),
                                ("tl_adagrad", "adagrad", 0.05, 3)):
        out.update(kat_case(name, 50, 40, 32, 64, "TL" if name.startswith("tl") else "BPR", opt, 1e-3, 1e-3, lr, ns, rng))
        names.append(name)
    out["names"] = np.array(names)
    try:
        kat_case("x", 10, 10, 8, 4, "BPR", "sparse_adam", 0.0, 0.0, 0.01, 1, rng)
        out["sparse_adam_error"] = np.array("")
    except RuntimeError as e:
        out["sparse_adam_error"] = np.array(str(e))
    np.savez_compressed(os.path.join(HERE, "kat_optimizers.npz"), **out)
    print("kat_optimizers.npz:", names, "| sparse_adam:", out["sparse_adam_error"])


# ----------------------------------------------------------------------------
class _TqdmCapture:
    """Stand-in for tqdm inside GeneralRecommender.fit (AbstractRecommender.py:116-129)
    that records the per-epoch `loss=` postfix."""
    epoch_losses = []

    def __init__(self, it):
        self.it = it

    def __iter__(self):
        return iter(self.it)

    def set_description(self, *_):
        pass

    def set_postfix(self, loss=None, **_):
        _TqdmCapture.epoch_losses.append(float(loss))


def make_ml100k(epochs=3, out="ml100k_c1.npz", factors=32, num_ng=1, **over):
    """`over`: config overrides (e.g. optimizer='adam', lr=0.001 -> ml100k_c1_adam.npz: the same run with
    torch.optim.Adam, AbstractRecommender.py:54).  factors=100, num_ng=4 -> ml100k_default.npz: what
    `python run_examples/test.py` runs when nothing is overridden (mf.yaml:1, basic.yaml:22-24)."""
    cwd = os.getcwd()
    os.chdir(REF)                        # data_path is relative ('data/'); nothing is written
    try:
        cfg = base_config(factors=factors, num_ng=num_ng, epochs=epochs, early_stop=False,
                          algo_name="mf", dataset="ml-100k", **over)
        seed_all(cfg["seed"])            # config.py:21-42 (CPU part)
        df = RawDataReader(cfg).get_data()
        pre = Preprocessor(cfg)
        df = pre.process(df)
        cfg["user_num"], cfg["item_num"] = pre.user_num, pre.item_num
        tr_idx, te_idx = TestSplitter(cfg).split(df)
        train_set, test_set = df.iloc[tr_idx, :].copy(), df.iloc[te_idx, :].copy()
        test_ur = get_ur(test_set)
        train_ur = get_ur(train_set)
        cfg["train_ur"] = train_ur

        model = MF(cfg)                                                   # test.py:90
        P0 = model.embed_user.weight.detach().numpy().copy()
        Q0 = model.embed_item.weight.detach().numpy().copy()
        train_users = train_set["user"].to_numpy().astype(np.int32)
        train_items = train_set["item"].to_numpy().astype(np.int32)
        samples = BasicNegtiveSampler(train_set, cfg).sampling()         # test.py:91-92
        loader = get_dataloader(BasicDataset(samples), batch_size=cfg["batch_size"],
                                shuffle=True, num_workers=0)              # test.py:93-94
        rng_state_before_fit = torch.get_rng_state().numpy().copy()
        ref_abs.tqdm = _TqdmCapture
        _TqdmCapture.epoch_losses = []
        model.fit(loader)                                                 # test.py:95
        epoch_losses = np.array(_TqdmCapture.epoch_losses, dtype=np.float64)
        P1 = model.embed_user.weight.detach().numpy().copy()
        Q1 = model.embed_item.weight.detach().numpy().copy()

        test_u, test_ucands = build_candidates_set(test_ur, train_ur, cfg)  # test.py:112
        cands = np.stack([c[1] for c in test_ucands]).astype(np.int64)
        test_loader = get_dataloader(CandidatesDataset(test_ucands), batch_size=128,
                                     shuffle=False, num_workers=0)
        preds = model.rank(test_loader)                                   # test.py:120
        full = np.stack([model.full_rank(int(u)) for u in test_u[:16]])
    finally:
        os.chdir(cwd)
    np.savez_compressed(
        os.path.join(HERE, out),
        user_num=np.int64(cfg["user_num"]), item_num=np.int64(cfg["item_num"]),
        hyper=np.array([cfg["lr"], cfg["reg_1"], cfg["reg_2"]], dtype=np.float64),
        factors=np.int64(factors), batch_size=np.int64(cfg["batch_size"]), epochs=np.int64(epochs),
        topk=np.int64(cfg["topk"]), seed=np.int64(cfg["seed"]),
        train_users=train_users, train_items=train_items,
        samples=samples.astype(np.int32), P0=P0, Q0=Q0,
        rng_state_before_fit=rng_state_before_fit,
        epoch_losses=epoch_losses, P1=P1, Q1=Q1,
        test_u=np.array(test_u, dtype=np.int64), cands=cands,
        preds=preds.astype(np.float32), full_rank16=full.astype(np.int64),
    )
    print(out, ": samples", samples.shape, "epoch losses", epoch_losses, "preds", preds.shape)


# ----------------------------------------------------------------------------
def make_rank_kat():
    rng = np.random.default_rng(7)
    U, I, d, C, nB, topk = 40, 300, 32, 100, 10, 10   # nB % 4 != 1: reference .squeeze() breaks on a 1-row tail batch (MFRecommender.py:115)
    cfg = base_config(user_num=U, item_num=I, factors=d, topk=topk)
    model = MF(cfg)
    P = (rng.standard_normal((U, d)) * 0.1).astype(np.float32)
    Q = (rng.standard_normal((I, d)) * 0.1).astype(np.float32)
    with torch.no_grad():
        model.embed_user.weight.copy_(torch.from_numpy(P))
        model.embed_item.weight.copy_(torch.from_numpy(Q))
    us = rng.integers(0, U, size=nB).astype(np.int64)
    cands = rng.integers(0, I, size=(nB, C)).astype(np.int64)   # with replacement -> duplicates
    ucands = [[int(us[b]), cands[b]] for b in range(nB)]
    loader = get_dataloader(CandidatesDataset(ucands), batch_size=4, shuffle=False, num_workers=0)
    preds = model.rank(loader)
    full = np.stack([model.full_rank(int(u)) for u in us])
    pred_pairs = np.array([model.predict(int(us[b]), int(cands[b, 0])) for b in range(nB)],
                          dtype=np.float32)
    np.savez_compressed(os.path.join(HERE, "rank_kat.npz"), P=P, Q=Q, us=us, cands=cands,
                        topk=np.int64(topk), preds=preds.astype(np.float32),
                        full=full.astype(np.int64), predict=pred_pairs)
    print("rank_kat.npz: preds", preds.shape, "full", full.shape)


if __name__ == "__main__":  # pragma: no cover
    make_kat_steps()
    make_rank_kat()
    make_ml100k()
    make_ml100k(out="ml100k_c1_adam.npz", optimizer="adam", lr=0.001)
    make_ml100k(epochs=2, out="ml100k_default.npz", factors=100, num_ng=4)
    make_kat_optimizers()
