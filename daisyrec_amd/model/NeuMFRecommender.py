"""NeuMF recommender with the reference's interface, trained by HIP kernels.

Mirror of daisy/model/NeuMFRecommender.py:15-233 (class ``NeuMF``): same config keys, the same
``nn.Module`` attributes (``embed_user_GMF`` ... ``MLP_layers`` ``predict_layer``, so ``state_dict()``
and NeuMF-pre style consumers keep working) created and initialised in the same order from the global
torch RNG, and the same methods.  ``fit`` / ``rank`` / ``full_rank`` / ``predict`` / ``calc_loss`` run
through ``daisy_neumf_*`` (include/daisyrec_amd.h): the MLP tower on fp32 MFMA tiles, gathers / loss
epilogue / embedding scatter as HBM-bound kernels, dense Adam (the model's default optimiser) or SGD
as one fused pass over all parameters.  There is no CPU path.

Dropout (neumf.yaml: 0.5) uses the device's counter-hash masks, not torch's generator: the same
distribution, a different stream; dropout = 0 reproduces the reference run.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

from .. import ops
from .. import _native as N
from .AbstractRecommender import GeneralRecommender, _tqdm


NEUMF_PRECISIONS = {"fp32": 0, "bf16_inputs": 1, "bf16": 2}     # daisy_neumf_ctx_set_precision levels


class NeuMF(GeneralRecommender):
    def __init__(self, config):
        """Config keys as in NeuMFRecommender.py:40-78."""
        super().__init__(config)
        self.lr = config["lr"]
        self.epochs = config["epochs"]
        self.reg_1 = config["reg_1"]
        self.reg_2 = config["reg_2"]
        self.dropout = config["dropout"]
        self.model = config["model_name"]
        self.GMF_model = config["GMF_model"]
        self.MLP_model = config["MLP_model"]
        self.factors = int(config["factors"])
        self.num_layers = int(config["num_layers"])
        # knob of the native path (absent from the reference config): 'fp32' = parity mode (default),
        # 'bf16' = bf16-input MFMA in the MLP tower (throughput mode, BASELINE configs[3])
        # 'fp32' (parity mode) / 'bf16_inputs' (fp32 in HBM, bf16 MFMA operands) / 'bf16' (bf16-stored activations)
        self.precision = str(config.get("precision", "fp32")).lower()
        if self.precision not in NEUMF_PRECISIONS:
            raise ValueError(f"config['precision'] must be one of {sorted(NEUMF_PRECISIONS)}, got {self.precision!r}")
        if self.model not in ("NeuMF-pre",) and self.model not in ops.NEUMF_MODELS:
            # the reference treats every other name as the full model (NeuMFRecommender.py:67-70,118-132)
            self._native_model = "NeuMF"
        else:
            self._native_model = "NeuMF" if self.model == "NeuMF-pre" else self.model

        dm = self.factors * (2 ** (self.num_layers - 1))
        self.embed_user_GMF = nn.Embedding(config["user_num"], self.factors)
        self.embed_item_GMF = nn.Embedding(config["item_num"], self.factors)
        self.embed_user_MLP = nn.Embedding(config["user_num"], dm)
        self.embed_item_MLP = nn.Embedding(config["item_num"], dm)
        mlp = []
        for i in range(self.num_layers):                                  # NeuMFRecommender.py:60-66
            n_in = self.factors * (2 ** (self.num_layers - i))
            mlp += [nn.Dropout(p=self.dropout), nn.Linear(n_in, n_in // 2), nn.ReLU()]
        self.MLP_layers = nn.Sequential(*mlp)
        predict_size = self.factors if self.model in ("MLP", "GMF") else self.factors * 2
        self.predict_layer = nn.Linear(predict_size, 1)

        self.loss_type = config["loss_type"]
        self.optimizer = config["optimizer"] if config["optimizer"] != "default" else "adam"
        self.initializer = config["init_method"] if config["init_method"] != "default" else "xavier_normal"
        self.early_stop = config["early_stop"]
        self.topk = config["topk"]
        self._init_weight()
        self._flat = None

    def _init_weight(self):
        """NeuMFRecommender.py:80-116 (note: the MLP Linear weights are initialised WITHOUT the
        per-initialiser keyword arguments, :90-92)."""
        init = self.initializer_config[self.initializer]
        kw = self.initializer_param_config[self.initializer]
        if self.model != "NeuMF-pre":
            for emb in (self.embed_user_GMF, self.embed_item_GMF, self.embed_user_MLP, self.embed_item_MLP):
                init(emb.weight, **kw)
            for m in self.MLP_layers:
                if isinstance(m, nn.Linear):
                    init(m.weight)
            init(self.predict_layer.weight, **kw)
            for m in self.modules():
                if isinstance(m, nn.Linear) and m.bias is not None:
                    m.bias.data.zero_()
        else:                                                             # :98-116, as written there
            self.embed_user_GMF.weight.data.copy_(self.GMF_model.embed_user_GMF.weight)
            self.embed_item_GMF.weight.data.copy_(self.GMF_model.embed_item_GMF.weight)
            self.embed_user_MLP.weight.data.copy_(self.MLP_model.embed_user_MLP.weight)
            self.embed_item_MLP.weight.data.copy_(self.MLP_model.embed_item_MLP.weight)
            for m1, m2 in zip(self.MLP_layers, self.MLP_model.MLP_layers):
                if isinstance(m1, nn.Linear) and isinstance(m2, nn.Linear):
                    m1.weight.data.copy_(m2.weight)
                    m1.bias.data.copy_(m2.bias)
            predict_weight = torch.cat([self.GMF_model.predict_layer.weight,
                                        self.MLP_model.predict_layer.weight], dim=1)
            predict_bias = self.GMF_model.predict_layer.bias + self.MLP_model.predict_layer.bias
            self.predict_layer.weight.data.copy_(0.5 * predict_weight)
            self.predict_layer.weight.data.copy_(0.5 * predict_bias)

    # -- parameters as the kernels see them: ONE flat device buffer, the module's tensors are views -----
    def _named(self):
        out = {"uG": self.embed_user_GMF.weight, "iG": self.embed_item_GMF.weight,
               "uM": self.embed_user_MLP.weight, "iM": self.embed_item_MLP.weight}
        lin = [m for m in self.MLP_layers if isinstance(m, nn.Linear)]
        for l, m in enumerate(lin, 1):
            out[f"W{l}"], out[f"b{l}"] = m.weight, m.bias
        out["Wp"], out["bp"] = self.predict_layer.weight, self.predict_layer.bias
        return out

    def _params(self):
        """Move the parameters into one contiguous device buffer (once) so that the optimiser is a
        single pass; returns dict name -> view."""
        self._require_device()
        named = self._named()
        if self._flat is None or not all(p.is_cuda for p in named.values()):
            total = sum(p.numel() for p in named.values())
            flat = torch.empty(total, dtype=torch.float32, device=self.device)
            off = 0
            for p in named.values():
                n = p.numel()
                flat[off:off + n].copy_(p.data.reshape(-1).to(flat.device))
                p.data = flat[off:off + n].view(p.shape)
                off += n
            self._flat = flat
        return {k: p.data for k, p in named.items()}

    def _views_like_flat(self, flat):
        out, off = {}, 0
        for k, p in self._named().items():
            n = p.numel()
            out[k] = flat[off:off + n].view(p.shape)
            off += n
        return out

    def _ctx(self, rows):
        ctx = ops.NeumfContext(rows, self.factors, self.num_layers, self.embed_user_GMF.num_embeddings,
                               self.embed_item_GMF.num_embeddings, model=self._native_model, device=self.device)
        ctx.set_precision(NEUMF_PRECISIONS[self.precision])
        return ctx

    # -- reference surface ---------------------------------------------------------------------------
    def forward(self, user, item):
        """NeuMFRecommender.py:118-137 (scored in eval mode)."""
        p = self._params()
        user = torch.as_tensor(user).to(self.device).reshape(-1)
        item = torch.as_tensor(item).to(self.device).reshape(-1)
        ctx = self._ctx(max(int(user.numel()), 1))
        try:
            return ctx.scores(p, user, item)
        finally:
            ctx.close()

    def calc_loss(self, batch):
        """NeuMFRecommender.py:139-169: the batch loss (0-dim float64 device tensor, no autograd graph)."""
        loss_id = self._build_criterion(self.loss_type)
        p = self._params()
        u, i, j = (torch.as_tensor(x).to(torch.int32).to(self.device).contiguous() for x in batch[:3])
        ctx = self._ctx(2 * u.numel())
        try:
            scratch = self._views_like_flat(torch.zeros_like(self._flat))
            ctx.step_grads(p, scratch, u, i, j, loss_id, self.reg_1, self.reg_2,
                           dropout=self.dropout if self.training else 0.0, seed=self.seed)
            return ctx.stats[N.NST_LOSS].clone()
        finally:
            ctx.close()

    def fit(self, train_loader):
        """AbstractRecommender.py:103-137 for NeuMF: per batch one `daisy_neumf_step_grads` and one
        optimiser pass over the flat parameter buffer; one host sync per epoch."""
        opt = self._resolve_optimizer()
        loss_id = self._build_criterion(self.loss_type)
        p = self._params()
        data = getattr(train_loader.dataset, "data", None)
        if data is None:
            raise TypeError("fit expects a DataLoader over BasicDataset (dataset.data = int32 [N,3] triples)")
        triples = torch.as_tensor(data).to(torch.int32).contiguous().to(self.device)
        n, B = triples.shape[0], int(train_loader.batch_size)
        if train_loader.drop_last:
            n = (n // B) * B
        gflat = torch.zeros_like(self._flat)
        grads = self._views_like_flat(gflat)
        optim = ops.DenseOptimizer(opt, self.lr)
        ctx = self._ctx(2 * min(B, max(n, 1)))
        self.epoch_losses, last_loss, step = [], 0.0, 0
        try:
            epochs = range(1, self.epochs + 1)
            bar = _tqdm(epochs) if (_tqdm is not None and self.show_progress) else None
            for epoch in (bar if bar is not None else epochs):
                self.train()
                perm = self._epoch_order(train_loader, triples.shape[0])
                order = triples[:n] if perm is None else triples[perm[:n].to(self.device)]
                # the epoch's ids column by column, once: a batch is then three views (at B = 256 the three per-batch copies
                # were 3 of the step's ~46 launches, each a few microseconds of latency)
                cols = [order[:, k].contiguous() for k in range(3)]
                # (the epoch's loss: every step adds its loss to stats[NST_LOSS_SUM] on the device - no launch per step for it)
                ctx.stats[N.NST_LOSS_SUM:N.NST_LOSS_SUM + 1].zero_()
                # the loop over the batches (zero_grad / calc_loss / backward / optimizer.step: step k uses the dropout seed
                # (self.seed << 32) | k) runs in the library: at 256 samples a step is ~40 us of kernels, less than the
                # Python of one iteration around two library calls
                step += ctx.fit_epoch(p, grads, cols[0], cols[1], cols[2], B, optim, self._flat, gflat, loss_id, self.reg_1,
                                      self.reg_2, dropout=self.dropout, seed_hi=(int(self.seed) & 0xFFFFFFFF) << 32, step0=step)
                current_loss = float(ctx.stats[N.NST_LOSS_SUM].cpu())
                if current_loss != current_loss or current_loss in (float("inf"), float("-inf")):
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

    def predict(self, u, i):
        """NeuMFRecommender.py:171-176."""
        return float(self.forward(torch.tensor([u]), torch.tensor([i])).cpu().item())

    def rank(self, test_loader):
        """NeuMFRecommender.py:178-209 -> float32 [n_users, topk] like the reference."""
        p = self._params()
        out, ctx = [], None
        try:
            for us, cands_ids in test_loader:
                us = torch.as_tensor(us).to(self.device).reshape(-1)
                cands_ids = torch.as_tensor(cands_ids).to(self.device)
                if cands_ids.dim() == 1:
                    cands_ids = cands_ids.unsqueeze(0)
                Bu, C = cands_ids.shape
                if ctx is None:
                    ctx = self._ctx(min(Bu * C, 1 << 18))
                scores = ctx.scores(p, us, cands_ids.reshape(-1), C_=C)
                out.append(ops.topk_from_scores(scores.view(Bu, C), cands_ids, self.topk))
        finally:
            if ctx is not None:
                ctx.close()
        if not out:
            return np.zeros((0,), dtype=np.float32)
        return torch.cat(out, 0).to(torch.float32).cpu().numpy()

    def full_rank(self, u):
        """NeuMFRecommender.py:211-233 -> int64 [topk]."""
        p = self._params()
        I = self.embed_item_GMF.num_embeddings
        ctx = self._ctx(min(I, 1 << 18))
        try:
            scores = ctx.scores(p, torch.tensor([int(u)], device=self.device), None, C_=0, n=I)
            return ops.full_topk_from_scores(scores, self.topk).cpu().numpy()
        finally:
            ctx.close()
