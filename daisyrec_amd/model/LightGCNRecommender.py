"""LightGCN recommender with the reference's interface, trained by HIP kernels.

Mirror of daisy/model/LightGCNRecommender.py:17-210 (class ``LightGCN``): same config keys
(``inter_matrix`` is the scipy COO matrix `utils.get_inter_matrix` puts into the config, test.py:88-89),
same attributes (``embed_user`` / ``embed_item``, ``restore_user_e`` / ``restore_item_e``) and methods.

One training step (``calc_loss`` + backward + optimiser step of the reference) is
    out  = mean_k A_hat^k [P; Q]                         daisy_lgcn_propagate   (2-3 sparse x dense products)
    loss, coefficients on rows of `out`                   daisy_bpr_forward / _finalize     (the MF kernels)
    G    = d loss / d out                                 daisy_bpr_item_grad_data + daisy_bpr_user_grad
    dE0  = 1/(L+1) sum_k A_hat^k G  (+ regulariser rows)  daisy_lgcn_backprop, daisy_lgcn_reg_grad
    Adam / SGD on the flat [P; Q] buffer                  daisy_adam_dense / daisy_sgd_dense
Like the reference, the propagation is recomputed for every batch.  There is no CPU path.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

from .. import ops
from .. import _native as N
from .AbstractRecommender import GeneralRecommender, _tqdm


class LightGCN(GeneralRecommender):
    def __init__(self, config):
        """Config keys as in LightGCNRecommender.py:39-62."""
        super().__init__(config)
        self.epochs = config["epochs"]
        self.lr = config["lr"]
        self.topk = config["topk"]
        self.user_num = config["user_num"]
        self.item_num = config["item_num"]
        self.interaction_matrix = config["inter_matrix"]
        self.factors = config["factors"]
        self.num_layers = config["num_layers"]
        self.reg_1 = config["reg_1"]
        self.reg_2 = config["reg_2"]

        self.embed_user = nn.Embedding(self.user_num, self.factors)
        self.embed_item = nn.Embedding(self.item_num, self.factors)

        self.loss_type = config["loss_type"]
        self.optimizer = config["optimizer"] if config["optimizer"] != "default" else "adam"
        self.initializer = config["init_method"] if config["init_method"] != "default" else "xavier_uniform"
        self.early_stop = config["early_stop"]

        # knob of the native path: 'chunked' (default here) = throughput kernels, results repeat to fp32
        # round-off; 'sorted' = bitwise reproducible run to run (row-owner products: every row's entries are
        # summed serially, which is slow on graphs with very popular items)
        self.item_mode = str(config.get("item_mode", "chunked")).lower()
        # multi-GPU (BASELINE configs[4]): 'auto' shards the propagation by node rows over the ranks of an
        # initialised torch.distributed job (every rank runs the same fit with the same seeds: the batches are
        # replicated, the sparse products - the bulk of a step - are split); False keeps everything local
        self.shard_rows = config.get("shard_rows", "auto")
        # multi-GPU propagation: sub-blocks per layer whose all-gather overlaps the next sub-block's reduction
        # (None: 4 on RCCL, 1 elsewhere)
        self.propagation_pieces = config.get("propagation_pieces")
        self._sharded = None
        self.restore_user_e = None
        self.restore_item_e = None
        self.apply(self._init_weight)
        self._flat = None
        self._graph = None

    # -- device state ------------------------------------------------------------------------------
    def _ego(self):
        """[P; Q] as ONE contiguous device buffer (get_ego_embeddings, :109-115, without the copy):
        the two nn.Embedding weights become views of it."""
        self._require_device()
        if self._flat is None or not self.embed_user.weight.is_cuda:
            U, I, d = self.user_num, self.item_num, self.factors
            flat = torch.empty((U + I) * d, dtype=torch.float32, device=self.device)
            flat[:U * d].copy_(self.embed_user.weight.data.reshape(-1))
            flat[U * d:].copy_(self.embed_item.weight.data.reshape(-1))
            self.embed_user.weight.data = flat[:U * d].view(U, d)
            self.embed_item.weight.data = flat[U * d:].view(I, d)
            self._flat = flat
        return self._flat.view(self.user_num + self.item_num, self.factors)

    def _adj(self):
        """get_norm_adj_mat (:74-107) on the device, built once."""
        if self._graph is None:
            m = self.interaction_matrix.tocoo()
            users = torch.as_tensor(np.ascontiguousarray(m.row)).to(self.device)
            items = torch.as_tensor(np.ascontiguousarray(m.col)).to(self.device)
            self._graph = ops.LgcnGraph(users, items, self.user_num, self.item_num)
            self._graph.set_reproducible(self.item_mode == "sorted")
        return self._graph

    def _prop(self):
        """the propagation engine: the local graph, or its row-sharded form over the job's ranks"""
        g = self._adj()
        import torch.distributed as dist
        want = self.shard_rows is True or (self.shard_rows == "auto" and dist.is_available() and dist.is_initialized()
                                           and dist.get_world_size() > 1)
        if not want:
            return g
        if self._sharded is None:
            from ..sharding import RowShardedPropagation
            self._sharded = RowShardedPropagation(g, self.user_num + self.item_num, self.factors, self.device,
                                                  pieces=self.propagation_pieces)
        return self._sharded

    def forward(self):
        """:117-129 -> (user_embedding [U,d], item_embedding [I,d])"""
        E0 = self._ego()
        out = self._prop().propagate(E0, self.num_layers, out=torch.empty_like(E0))
        return out[:self.user_num], out[self.user_num:]

    def _batch_grads(self, ctx, E0, out, G, dE0, u, i, j, loss_id):
        """loss (left in ctx.stats) and d loss / d E0 accumulated into dE0 for one batch."""
        U = self.user_num
        reg = self.reg_1 != 0 or self.reg_2 != 0
        pointwise = loss_id in ops.POINTWISE_LOSSES
        ctx.set_batch(u, i, j)
        self._prop().propagate(E0, self.num_layers, out=out)
        if reg:                       # the regularisers act on the EGO rows (:150-163): their sums first
            ctx.forward(E0[:U], E0[U:], loss_id)
            ego = ctx.stats[1:7].clone()
        ctx.forward(out[:U], out[U:], loss_id)
        if reg:
            ctx.stats[1:7] = ego
        else:
            ctx.stats[1:7] = 0
        ctx.finalize(self.reg_1, self.reg_2, accumulate=True)
        G.zero_()
        if self.item_mode == "sorted":       # owner-summed in plan order; reg 0: the data term only
            ctx.item_grad(out[:U], out[U:], 0.0, 0.0, N.ITEM_SORTED, gQ=G[U:])
        else:
            ctx.item_grad_data(out[:U], out[U:], N.ITEM_CHUNKED, gQ=G[U:])
        ctx.user_grad(out[:U], out[U:], 0.0, 0.0, G[:U])
        self._prop().backprop(G, self.num_layers, dE0)
        if reg:
            ops.lgcn_reg_grad(E0, u, i, j, U, pointwise, self.reg_1, self.reg_2, ctx.stats, dE0)

    def calc_loss(self, batch):
        """:131-169: the batch loss (0-dim float64 device tensor; no autograd graph)."""
        loss_id = self._build_criterion(self.loss_type)
        self.restore_user_e, self.restore_item_e = None, None
        E0 = self._ego()
        u, i, j = (torch.as_tensor(x).to(torch.int32).to(self.device).contiguous() for x in batch[:3])
        ctx = ops.BprContext(u.numel(), self.factors, self.user_num, self.item_num, device=self.device)
        try:
            ctx.set_pointwise(loss_id in ops.POINTWISE_LOSSES)
            out, G, dE0 = torch.empty_like(E0), torch.empty_like(E0), torch.zeros_like(E0)
            self._batch_grads(ctx, E0, out, G, dE0, u, i, j, loss_id)
            return ctx.stats[N.ST_LOSS].clone()
        finally:
            ctx.close()

    def fit(self, train_loader):
        """AbstractRecommender.py:103-137 for LightGCN."""
        opt = self._resolve_optimizer()
        loss_id = self._build_criterion(self.loss_type)
        E0 = self._ego()
        self.restore_user_e, self.restore_item_e = None, None
        data = getattr(train_loader.dataset, "data", None)
        if data is None:
            raise TypeError("fit expects a DataLoader over BasicDataset (dataset.data = int32 [N,3] triples)")
        triples = torch.as_tensor(data).to(torch.int32).contiguous().to(self.device)
        n, B = triples.shape[0], int(train_loader.batch_size)
        if train_loader.drop_last:
            n = (n // B) * B
        out, G, dE0 = torch.empty_like(E0), torch.empty_like(E0), torch.zeros_like(E0)
        gflat = dE0.view(-1)
        optim = ops.DenseOptimizer(opt, self.lr)
        ctx = ops.BprContext(min(B, max(n, 1)), self.factors, self.user_num, self.item_num, device=self.device)
        ctx.set_pointwise(loss_id in ops.POINTWISE_LOSSES)
        self.epoch_losses, last_loss, step = [], 0.0, 0
        try:
            epochs = range(1, self.epochs + 1)
            bar = _tqdm(epochs) if (_tqdm is not None and self.show_progress) else None
            for epoch in (bar if bar is not None else epochs):
                self.train()
                perm = self._epoch_order(train_loader, triples.shape[0])
                order = triples[:n] if perm is None else triples[perm[:n].to(self.device)]
                ctx.epoch_acc.zero_()
                for s in range(0, n, B):
                    rows = order[s:s + B]
                    u, i, j = (rows[:, k].contiguous() for k in range(3))
                    step += 1
                    self._batch_grads(ctx, E0, out, G, dE0, u, i, j, loss_id)
                    optim.next_step()
                    optim.step(self._flat, gflat)          # also clears the gradient
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

    def _restore(self):
        if self.restore_user_e is None or self.restore_item_e is None:       # :172-173,179-180,203-204
            self.restore_user_e, self.restore_item_e = self.forward()
        return self.restore_user_e, self.restore_item_e

    def predict(self, u, i):
        """:171-176"""
        ue, ie = self._restore()
        return float(ops.mf_predict(ue, ie, torch.tensor([u], device=self.device),
                                    torch.tensor([i], device=self.device)).cpu().item())

    def rank(self, test_loader):
        """:178-200 -> float32 [n_users, topk] like the reference."""
        ue, ie = self._restore()
        out = []
        for us, cands_ids in test_loader:
            us = torch.as_tensor(us).to(self.device)
            cands_ids = torch.as_tensor(cands_ids).to(self.device)
            if cands_ids.dim() == 1:
                cands_ids = cands_ids.unsqueeze(0)
            out.append(ops.mf_rank_topk(ue, ie, us.reshape(-1), cands_ids, self.topk))
        if not out:
            return np.zeros((0,), dtype=np.float32)
        return torch.cat(out, 0).to(torch.float32).cpu().numpy()

    def full_rank(self, u):
        """:202-210 -> int64 [topk]"""
        ue, ie = self._restore()
        return ops.mf_full_rank(ue, ie, int(u), self.topk).cpu().numpy()
