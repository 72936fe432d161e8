"""MF recommender with the reference's interface, trained by HIP kernels.

Mirror of daisy/model/MFRecommender.py:25-133 (class ``MF``): same config keys,
attributes (``embed_user`` / ``embed_item`` are real ``nn.Embedding`` modules
whose ``.weight`` storage the kernels update in place, so ``state_dict()`` and
NeuMF-style consumers of ``.embed_*.weight`` keep working) and methods.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

from .. import ops
from .AbstractRecommender import GeneralRecommender


class MF(GeneralRecommender):
    def __init__(self, config):
        """Config keys as in MFRecommender.py:44-59:
        user_num, item_num, factors, epochs, lr, reg_1, reg_2, topk, loss_type,
        optimizer, init_method, early_stop, gpu, logger."""
        super().__init__(config)
        self.lr = config["lr"]
        self.reg_1 = config["reg_1"]
        self.reg_2 = config["reg_2"]
        self.epochs = config["epochs"]
        self.topk = config["topk"]

        # created and initialised on the host from the global torch RNG, user table
        # first, exactly like MFRecommender.py:53-61; moved to the GPU by fit()
        self.embed_user = nn.Embedding(config["user_num"], config["factors"])
        self.embed_item = nn.Embedding(config["item_num"], config["factors"])

        self.loss_type = config["loss_type"]
        self.optimizer = config["optimizer"] if config["optimizer"] != "default" else "sgd"
        self.initializer = config["init_method"] if config["init_method"] != "default" else "normal"
        self.early_stop = config["early_stop"]

        self.apply(self._init_weight)

    # -- tables as the kernels see them ------------------------------------------
    def _tables(self):
        self._require_device()
        if not self.embed_user.weight.is_cuda:
            self.to(self.device)
        return self.embed_user.weight.data, self.embed_item.weight.data

    def _biases(self):
        """None for MF; FM returns (u_bias, i_bias, bias_) device tensors (FMRecommender.py:49-53)."""
        return None

    def forward(self, user, item):
        """MFRecommender.py:63-68: pred = (P[user] * Q[item]).sum(-1)."""
        P, Q = self._tables()
        user = torch.as_tensor(user).to(P.device)
        item = torch.as_tensor(item).to(P.device)
        return ops.mf_predict(P, Q, user.reshape(-1), item.reshape(-1), biases=self._biases()).view(user.shape)

    def calc_loss(self, batch):
        """MFRecommender.py:70-97 for the pairwise losses: returns the batch loss
        (0-dim float64 tensor on the device; no autograd graph — the gradient is
        produced by the update kernels, not by backward())."""
        loss_id = self._build_criterion(self.loss_type)          # raises on invalid types
        P, Q = self._tables()
        u, i, j = (torch.as_tensor(x).to(torch.int32).to(P.device).contiguous() for x in batch[:3])
        ctx = ops.BprContext(u.numel(), P.shape[1], P.shape[0], Q.shape[0], device=P.device)
        try:
            ctx.set_pointwise(loss_id in ops.POINTWISE_LOSSES)   # batch[2] is the label (MFRecommender.py:76)
            b = self._biases()
            if b is not None:
                ctx.set_bias(*b, g_i_bias=torch.zeros(Q.shape[0], device=P.device))
            ctx.set_batch(u, i, j)
            ctx.forward(P, Q, loss_id)
            out = torch.zeros((), dtype=torch.float64, device=P.device)
            ctx.finalize(self.reg_1, self.reg_2, step_loss=out.view(1), accumulate=False)
            torch.cuda.synchronize()
        finally:
            ctx.close()
        return out

    def predict(self, u, i):
        """MFRecommender.py:99-104."""
        P, Q = self._tables()
        u = torch.tensor([u], device=P.device)
        i = torch.tensor([i], device=P.device)
        return float(ops.mf_predict(P, Q, u, i, biases=self._biases()).cpu().item())

    def rank(self, test_loader):
        """MFRecommender.py:106-123.  Returns float32 [n_users, topk] like the reference
        (ids are concatenated onto a float tensor there, MFRecommender.py:107,121)."""
        P, Q = self._tables()
        out = []
        for us, cands_ids in test_loader:
            us = torch.as_tensor(us).to(P.device)
            cands_ids = torch.as_tensor(cands_ids).to(P.device)
            if cands_ids.dim() == 1:
                cands_ids = cands_ids.unsqueeze(0)
            out.append(ops.mf_rank_topk(P, Q, us.reshape(-1), cands_ids, self.topk, biases=self._biases()))
        if not out:
            return np.zeros((0,), dtype=np.float32)
        return torch.cat(out, 0).to(torch.float32).cpu().numpy()

    def full_rank(self, u):
        """MFRecommender.py:126-133 -> int64 [topk]."""
        P, Q = self._tables()
        return ops.mf_full_rank(P, Q, int(u), self.topk, biases=self._biases()).cpu().numpy()
