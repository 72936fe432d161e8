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


def padded_factors(d: int, row_pitch="auto", batch=None) -> int:
    """columns of the tables the kernels train on for a d-factor model.  'auto': the next multiple of 32 (128-byte rows)
    where that was measured to pay - up to 64 columns and at most a third more of them (d = 24 -> 32: 0.433 -> 0.402 ms per
    2 M-sample step, d = 50 -> 64: 0.698 -> 0.603; but d = 100 -> 128: 1.131 -> 1.203, d = 200 -> 256: 2.56 -> 2.69 -
    profiles/r03_factor_sweep.txt); for batches of the reference's default size (<= 256, basic.yaml:23), where a step is a
    chain of dependent phases and not bytes, also beyond 64 columns: the default d = 100 runs as 128 (21.3 -> 17.3 us per
    step, profiles/r02_small_epoch.txt).  0 / False: never; an integer: pad to its next multiple."""
    if row_pitch in (0, False, None, "0", "off", "none"):
        return d
    if row_pitch == "auto":
        dp = (d + 31) // 32 * 32
        small = batch is not None and batch <= ops.SMALL_BATCH_MAX
        return dp if ((dp <= 64 or (small and dp <= 128)) and 4 * d >= 3 * dp) else d
    m = int(row_pitch)
    return (d + m - 1) // m * m


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
        self.row_pitch = config.get("row_pitch", "auto")
        self._padded = {}
        # a table homed on a row pitch is an [n, d] VIEW of a padded buffer: torch.save(state_dict()) would write the
        # whole padded storage (up to a third more bytes, and not byte-comparable with the reference's checkpoint), so
        # the state dict carries contiguous [n, d] copies of such tables.  Consequences a caller can see: for
        # factor counts that pad (d = 50, 100, ...) the entries of state_dict() do NOT alias the parameters - writing into
        # model.state_dict()[k] in place does not change the model there (load_state_dict does, as always) - and every
        # state_dict() call allocates the two copies on the device; factor counts that are whole 128-byte lines (32, 64,
        # 128) keep torch's aliasing entries.  row_pitch=0 switches the padding - and with it this difference - off.
        self._register_state_dict_hook(MF._contiguous_state)

        self.apply(self._init_weight)

    @staticmethod
    def _contiguous_state(module, state_dict, prefix, local_metadata):
        for name in ("embed_user.weight", "embed_item.weight"):
            t = state_dict.get(prefix + name)
            if t is not None and not t.is_contiguous():
                state_dict[prefix + name] = t.contiguous()
        return state_dict

    # -- tables as the kernels see them ------------------------------------------
    def _tables(self, batch=None):
        """(P, Q) as the kernels see them.  Row pitch (config['row_pitch'], default 'auto'): a factor count whose rows
        are not whole 128-byte lines (d = 50: 200-byte rows straddle lines and leave 3 of 16 lanes idle) trains on tables
        PADDED to the next multiple of 32 columns, the extra columns zero: they stay exactly zero under every update
        (their data gradient is c * 0, sign(0) = 0, the Frobenius and Adam terms are multiples of the element), so scores,
        norms, losses and gradients are those of the d-column model, while a row costs what the aligned row costs
        (profiles/r04_factor_sweep.txt).  `embed_*.weight` stays an [n, d] tensor - a strided view of the padded buffer -
        for state_dict() and every consumer of the reference's attribute.  `batch`: the batch size of the fit about to
        run (the pitch that pays depends on it); None keeps the tables as they are homed."""
        self._require_device()
        if not self.embed_user.weight.is_cuda:
            self.to(self.device)
        d = self.embed_user.weight.shape[1]
        want = padded_factors(d, self.row_pitch, batch)
        out = []
        for name in ("embed_user", "embed_item"):
            w = getattr(self, name).weight
            buf = self._padded.get(name)
            homed = (buf is not None and buf.device == w.device and w.data.data_ptr() == buf.data_ptr()
                     and tuple(w.data.stride()) == (buf.shape[1], 1))
            if homed and (batch is None or buf.shape[1] == want):
                out.append(buf)
            elif want == d:                                     # bare rows: the weight itself, contiguous
                if homed or not w.data.is_contiguous():
                    w.data = w.data.contiguous()
                self._padded.pop(name, None)
                out.append(w.data)
            else:                                               # (re)home the weights: after the first .to(device), a
                buf = torch.zeros(w.shape[0], want, dtype=w.dtype, device=w.device)    # load, another batch regime, ...
                buf[:, :d].copy_(w.data)
                w.data = buf[:, :d]
                self._padded[name] = buf
                out.append(buf)
        return out[0], out[1]

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
