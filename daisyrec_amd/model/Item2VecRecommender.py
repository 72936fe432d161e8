"""Item2Vec recommender with the reference's interface, trained by the MF kernels.

Mirror of daisy/model/Item2VecRecommender.py:15-112: one shared item table, skip-gram pairs
(target, context, label) from `SkipGramNegativeSampler`, BCE-with-logits (sum), Adam by default; after
training the user table is the sum of each user's training items' vectors.  A training step is the
point-wise MF path with P = Q = S (include/daisyrec_amd.h, Item2Vec section); there is no CPU path.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

from .. import ops
from .. import _native as N
from .AbstractRecommender import GeneralRecommender, _tqdm


class Item2Vec(GeneralRecommender):
    def __init__(self, config):
        """Config keys as in Item2VecRecommender.py:29-45: user_num, item_num, factors, lr, epochs, train_ur,
        optimizer, init_method, early_stop, topk, gpu, logger."""
        super().__init__(config)
        self.user_embedding = nn.Embedding(config["user_num"], config["factors"])
        self.ur = config["train_ur"]
        self.shared_embedding = nn.Embedding(config["item_num"], config["factors"])
        self.lr = config["lr"]
        self.epochs = config["epochs"]
        self.loss_type = "CL"                                                     # :38-39
        self.optimizer = config["optimizer"] if config["optimizer"] != "default" else "adam"
        self.initializer = config["init_method"] if config["init_method"] != "default" else "normal"
        self.early_stop = config["early_stop"]
        self.topk = config["topk"]
        self.apply(self._init_weight)

    def _tables(self):
        self._require_device()
        if not self.shared_embedding.weight.is_cuda:
            self.to(self.device)
        return self.user_embedding.weight.data, self.shared_embedding.weight.data

    def forward(self, target_i, context_j):
        """:47-51"""
        _, S = self._tables()
        t = torch.as_tensor(target_i).to(S.device).reshape(-1)
        c = torch.as_tensor(context_j).to(S.device).reshape(-1)
        return ops.mf_predict(S, S, t, c)

    def _step(self, ctx, S, gA, gB, t, c, y):
        """loss into ctx.stats / epoch_acc; d loss / d S into gA (gB is scratch, left zero)."""
        ctx.set_batch(t, c, y)
        ctx.forward(S, S, N.LOSS_CL)
        ctx.stats[1:7] = 0                          # no regulariser in Item2Vec.calc_loss (:61-69)
        ctx.finalize(0.0, 0.0, accumulate=True)
        ctx.item_grad_data(S, S, N.ITEM_CHUNKED, gQ=gB)        # rows of the contexts
        ctx.user_grad(S, S, 0.0, 0.0, gA)                      # rows of the targets
        ops.axpby(gB, 1.0, 1.0, gA, zero_x=True)

    def calc_loss(self, batch):
        """:61-69 (0-dim float64 device tensor; no autograd graph)."""
        _, S = self._tables()
        t, c, y = (torch.as_tensor(x).to(torch.int32).to(S.device).contiguous() for x in batch[:3])
        ctx = ops.BprContext(t.numel(), S.shape[1], S.shape[0], S.shape[0], device=S.device)
        try:
            ctx.set_pointwise(True)
            self._step(ctx, S, torch.zeros_like(S), torch.zeros_like(S), t, c, y)
            return ctx.stats[N.ST_LOSS].clone()
        finally:
            ctx.close()

    def fit(self, train_loader):
        """AbstractRecommender.py:103-137, then the user-embedding build of Item2VecRecommender.py:53-59."""
        opt = self._resolve_optimizer()
        Uemb, S = self._tables()
        data = getattr(train_loader.dataset, "data", None)
        if data is None:
            raise TypeError("fit expects a DataLoader over BasicDataset (dataset.data = int [N,3] triples)")
        triples = torch.as_tensor(np.asarray(data)).to(torch.int32).contiguous().to(S.device)
        n, B = triples.shape[0], int(train_loader.batch_size)
        if train_loader.drop_last:
            n = (n // B) * B
        gA, gB = torch.zeros_like(S), torch.zeros_like(S)
        optim = ops.DenseOptimizer(opt, self.lr)
        ctx = ops.BprContext(min(B, max(n, 1)), S.shape[1], S.shape[0], S.shape[0], device=S.device)
        ctx.set_pointwise(True)
        self.epoch_losses, last_loss, step = [], 0.0, 0
        try:
            epochs = range(1, self.epochs + 1)
            bar = _tqdm(epochs) if (_tqdm is not None and self.show_progress) else None
            for epoch in (bar if bar is not None else epochs):
                self.train()
                perm = self._epoch_order(train_loader, triples.shape[0])
                order = triples[:n] if perm is None else triples[perm[:n].to(S.device)]
                ctx.epoch_acc.zero_()
                for s in range(0, n, B):
                    rows = order[s:s + B]
                    t, c, y = (rows[:, k].contiguous() for k in range(3))
                    step += 1
                    self._step(ctx, S, gA, gB, t, c, y)
                    optim.next_step()
                    optim.step(S, gA)          # also clears the gradient
                acc = ctx.epoch_acc.cpu()
                current_loss = float(acc[0])
                if float(acc[1]) > 0 or current_loss != current_loss:
                    raise ValueError("Loss=Nan or Infinity: current settings does not fit the recommender")
                self.epoch_losses.append(current_loss)
                if bar is not None:
                    bar.set_description(f"[Epoch {epoch:03d}]")
                    bar.set_postfix(loss=current_loss)
                self.eval()
                if abs(current_loss - last_loss) < 1e-5 and self.early_stop:
                    self.logger.info("Satisfy early stop mechanism")
                    break
                last_loss = current_loss
        finally:
            torch.cuda.synchronize()
            ctx.close()
        self.logger.info("Start building user embedding...")
        users = np.concatenate([np.full(len(v), u, np.int32) for u, v in self.ur.items()]) if self.ur else np.zeros(0, np.int32)
        items = np.concatenate([np.fromiter(v, np.int32, len(v)) for v in self.ur.values()]) if self.ur else np.zeros(0, np.int32)
        if users.size:
            indptr, cols = ops.build_user_csr(torch.from_numpy(users).to(S.device), torch.from_numpy(items).to(S.device),
                                              Uemb.shape[0])
            ops.csr_row_sum(indptr, cols, S, Uemb)

    def predict(self, u, i):
        """:71-80"""
        Uemb, S = self._tables()
        return float(ops.mf_predict(Uemb, S, torch.tensor([u], device=S.device), torch.tensor([i], device=S.device)).cpu().item())

    def rank(self, test_loader):
        """:82-101 -> float32 [n_users, topk]"""
        Uemb, S = self._tables()
        out = []
        for us, cands_ids in test_loader:
            us = torch.as_tensor(us).to(S.device)
            cands_ids = torch.as_tensor(cands_ids).to(S.device)
            if cands_ids.dim() == 1:
                cands_ids = cands_ids.unsqueeze(0)
            out.append(ops.mf_rank_topk(Uemb, S, us.reshape(-1), cands_ids, self.topk))
        if not out:
            return np.zeros((0,), dtype=np.float32)
        return torch.cat(out, 0).to(torch.float32).cpu().numpy()

    def full_rank(self, u):
        """:103-112 -> int64 [topk]"""
        Uemb, S = self._tables()
        return ops.mf_full_rank(Uemb, S, int(u), self.topk).cpu().numpy()
