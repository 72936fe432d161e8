#!/usr/bin/env python
"""Golden vectors for NeuMF at the BASELINE configs[3] tower shape (factors 64, 3 layers: 512 -> 256 -> 128 -> 64), generated
from the REAL reference (`daisy.model.NeuMFRecommender.NeuMF`, imported from /root/reference; nothing is copied) on ml-100k in
run_examples/test.py's call order.  Runs only in the build container; the output tests/golden/kat_neumf_d64.npz is committed.

    python tests/golden/make_golden_neumf_d64.py

Why a second file: the reference's default NeuMF (factors 24, 2 layers: tests/golden/kat_neumf.npz) has no layer width that
tiles for the bf16-storage kernels, so a fit at that shape says nothing about the bf16 mode.  Here every width is a multiple
of 64 and the batch (2048 samples = 4096 rows per step) has more rows than the two tables together (943 + 1152), so the HIP
path runs the first layer through the tables and the fused tower kernel (csrc/neumf_tower.hip) in its bf16 mode - and the
same fit in its fp32 mode.  SGD, dropout 0, one epoch over the first 24 576 triples (12 batches).  lr = 2e-5: the criterion
is a SUM over the batch (AbstractRecommender.py:79-93), so at 2048 samples per batch the reference's lr 0.001 is an effective
step of ~2 on the mean loss - the fit is then chaotic (measured: two bf16 implementations that agree to 6e-4 on the first
step's gradient are 14 % apart on the third step's and O(1) apart on the sixth's; fp32 survives 12 steps only because it
starts from 1e-7).  At 2e-5 twelve steps stay in the regime where a difference stays the size it started with.  To keep the fixture small the tables are stored as a checksum of the initial state (the
test re-creates it from the seed and compares) and 48 rows each of the final state; the MLP and predict parameters in full.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G  # noqa: E402
import make_golden_neumf as GN  # noqa: E402

import torch  # noqa: E402
import daisy.model.AbstractRecommender as ref_abs  # noqa: E402
from daisy.model.NeuMFRecommender import NeuMF  # noqa: E402
from daisy.utils.dataset import BasicDataset, CandidatesDataset, get_dataloader  # noqa: E402
from daisy.utils.loader import Preprocessor, RawDataReader  # noqa: E402
from daisy.utils.sampler import BasicNegtiveSampler  # noqa: E402
from daisy.utils.splitter import TestSplitter  # noqa: E402
from daisy.utils.utils import build_candidates_set, get_ur  # noqa: E402

ROWS = 48            # rows of each embedding table kept from the final state


def main():
    cwd = os.getcwd()
    os.chdir(G.REF)
    try:
        cfg = GN.neumf_config(num_ng=1, epochs=1, early_stop=False, algo_name="neumf", dataset="ml-100k", factors=64,
                              num_layers=3, batch_size=2048, optimizer="sgd", lr=2e-5)
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
        init = {k: p.detach().numpy().copy() for k, p in GN.named_params(model).items()}
        samples = BasicNegtiveSampler(train_set, cfg).sampling()[:12 * 2048]
        loader = get_dataloader(BasicDataset(samples), batch_size=cfg["batch_size"], shuffle=True, num_workers=0)
        rng_state = torch.get_rng_state().numpy().copy()
        ref_abs.tqdm = G._TqdmCapture
        G._TqdmCapture.epoch_losses = []
        model.fit(loader)
        epoch_losses = np.array(G._TqdmCapture.epoch_losses, dtype=np.float64)
        final = {k: p.detach().numpy().copy() for k, p in GN.named_params(model).items()}
        test_u, test_ucands = build_candidates_set(test_ur, train_ur, cfg)
        cands = np.stack([c[1] for c in test_ucands]).astype(np.int64)
        preds = model.rank(get_dataloader(CandidatesDataset(test_ucands), batch_size=128, shuffle=False, num_workers=0))
    finally:
        os.chdir(cwd)
    out = {"meta": np.array([cfg["user_num"], cfg["item_num"], cfg["factors"], cfg["num_layers"]], dtype=np.int64),
           "hyper": np.array([cfg["lr"], cfg["reg_1"], cfg["reg_2"]], dtype=np.float64), "optimizer": np.array(model.optimizer),
           "batch_size": np.int64(cfg["batch_size"]), "epochs": np.int64(1), "topk": np.int64(cfg["topk"]),
           "seed": np.int64(cfg["seed"]), "samples": samples.astype(np.int32), "rng_state_before_fit": rng_state,
           "epoch_losses": epoch_losses, "test_u": np.array(test_u[:64], dtype=np.int64), "cands": cands[:64],
           "preds": preds[:64].astype(np.float32)}
    for k in init:
        out[f"{k}0_sum"] = np.array([init[k].astype(np.float64).sum(), np.abs(init[k].astype(np.float64)).sum()])
        out[f"{k}1_norm"] = np.float64(np.linalg.norm(final[k].astype(np.float64)))
        out[f"{k}_delta_norm"] = np.float64(np.linalg.norm((final[k] - init[k]).astype(np.float64)))
        out[f"{k}1"] = final[k][:ROWS] if k in ("uG", "iG", "uM", "iM") else final[k]
        # the step the fit took, on the kept rows / the whole small tensor: what a wrong gradient would show up in
        out[f"{k}_delta"] = (final[k] - init[k])[:ROWS] if k in ("uG", "iG", "uM", "iM") else (final[k] - init[k])
    print("ml-100k NeuMF d=64 L=3 (sgd): samples", samples.shape, "epoch losses", epoch_losses, "preds", preds.shape)
    np.savez_compressed(os.path.join(HERE, "kat_neumf_d64.npz"), **out)
    print("kat_neumf_d64.npz:", os.path.getsize(os.path.join(HERE, "kat_neumf_d64.npz")), "bytes")


if __name__ == "__main__":
    main()
