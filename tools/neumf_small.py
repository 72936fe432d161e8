#!/usr/bin/env python
"""NeuMF at the reference's own operating point (neumf.yaml: factors 24, 2 layers, dropout 0.5, Adam; basic.yaml:23: batch 256) on
ml-100k sizes, through NeuMF.fit: us per step (python tools/neumf_small.py [factors] [layers] [batch] [dropout])"""
import logging
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from daisyrec_amd.model.NeuMFRecommender import NeuMF  # noqa: E402
from daisyrec_amd.utils.dataset import BasicDataset, get_dataloader  # noqa: E402

d = int(sys.argv[1]) if len(sys.argv) > 1 else 24
L = int(sys.argv[2]) if len(sys.argv) > 2 else 2
B = int(sys.argv[3]) if len(sys.argv) > 3 else 256
drop = float(sys.argv[4]) if len(sys.argv) > 4 else 0.5
U, I, n = 943, 1152, 256 * 300
rng = np.random.default_rng(0)
tri = np.stack([rng.integers(0, U, n), rng.integers(0, I, n), rng.integers(0, I, n)], 1).astype(np.int32)
cfg = {"gpu": "0", "logger": logging.getLogger("t"), "lr": 0.001, "reg_1": 0.0, "reg_2": 0.001, "epochs": 1, "topk": 50,
       "user_num": U, "item_num": I, "factors": d, "num_layers": L, "dropout": drop, "loss_type": "BPR", "optimizer": "adam",
       "init_method": "default", "early_stop": False, "model_name": "NeuMF", "GMF_model": None, "MLP_model": None,
       "algo_name": "neumf", "progress": False}
model = NeuMF(cfg)
loader = get_dataloader(BasicDataset(tri), batch_size=B, shuffle=False, num_workers=0)
model.fit(loader)
torch.cuda.synchronize()
t0 = time.perf_counter()
model.fit(loader)
torch.cuda.synchronize()
steps = (n + B - 1) // B
print(f"NeuMF factors={d} layers={L} B={B} dropout={drop} Adam, ml-100k sizes: {(time.perf_counter() - t0) / steps * 1e6:.1f} us per step "
      f"({steps} steps, epoch loss {model.epoch_losses[-1]:.3f})", flush=True)
