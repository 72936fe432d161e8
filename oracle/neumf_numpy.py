"""CPU restatement of the reference's NeuMF recommender — TEST INFRASTRUCTURE ONLY.

Follows `daisy/model/NeuMFRecommender.py`:
  forward   (:118-137)  GMF branch  uG[u] * iG[item];  MLP branch  [uM[u] | iM[item]] through
                        num_layers x (Dropout -> Linear(n, n/2) -> ReLU);  predict_layer on the
                        concatenation (model 'GMF' / 'MLP' use one branch only)
  calc_loss (:139-169)  criterion on (pos, neg) [BPR/HL/TL] or (pos, label) [CL/SL], sum reduction
                        (AbstractRecommender.py:79-93), plus the non-squared regularisers EXACTLY
                        as the reference writes them — including its quirk that the negative item's
                        GMF rows are counted twice and its MLP rows never (:158-161)
  rank / full_rank / predict (:171-233)
Dropout: the oracle takes the keep masks as inputs (`masks[l]`, already scaled by 1/(1-p)); the
reference draws them from torch's global generator, which no counter-based device generator can
reproduce, so golden vectors are generated at dropout=0 and the mask path is checked HIP-vs-oracle
with the device's own Philox masks.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this module.
Pinned by tests/golden/kat_neumf.npz (tests/golden/make_golden_neumf.py).
"""
import numpy as np

from .bpr_mf_numpy import LOSS_CL, LOSS_SL, LOSS_BPR, _mix32, _sigmoid, pair_loss_coef  # noqa: F401

PARAM_ORDER = ("uG", "iG", "uM", "iM")          # embedding tables; then W1,b1,...,WL,bL, Wp, bp


def param_names(num_layers):
    names = list(PARAM_ORDER)
    for l in range(1, num_layers + 1):
        names += [f"W{l}", f"b{l}"]
    return names + ["Wp", "bp"]


def _mlp_forward(x0, p, num_layers, masks=None):
    """Returns the list [x0_dropped, x1, ..., xL] of layer inputs/outputs (post-ReLU)."""
    acts, drops = [], []
    x = x0
    for l in range(1, num_layers + 1):
        xd = x * masks[l - 1] if masks is not None else x
        drops.append(xd)
        x = np.maximum(xd @ p[f"W{l}"].T + p[f"b{l}"], 0.0)
        acts.append(x)
    return drops, acts


def neumf_forward(p, u, item, num_layers, model="NeuMF", masks=None, dtype=np.float32):
    """NeuMFRecommender.py:118-137 -> (pred [R], cache for the backward pass)."""
    p = {k: np.asarray(v, dtype) for k, v in p.items()}
    u = np.asarray(u, np.int64)
    item = np.asarray(item, np.int64)
    cache = {"u": u, "item": item}
    parts = []
    if model != "MLP":
        cache["g"] = p["uG"][u] * p["iG"][item]
        parts.append(cache["g"])
    if model != "GMF":
        x0 = np.concatenate([p["uM"][u], p["iM"][item]], axis=-1)
        cache["drops"], cache["acts"] = _mlp_forward(x0, p, num_layers, masks)
        parts.append(cache["acts"][-1])
    concat = np.concatenate(parts, axis=-1)
    cache["concat"] = concat
    pred = concat @ p["Wp"].reshape(-1) + p["bp"].reshape(-1)[0]
    return pred, cache


def _backward_rows(p, cache, dpred, num_layers, model, masks, grads):
    """Accumulate d loss / d params for one forward(user, item) call given d loss / d pred."""
    u, item = cache["u"], cache["item"]
    wp = p["Wp"].reshape(-1)
    grads["Wp"] += (dpred[:, None] * cache["concat"]).sum(0).reshape(p["Wp"].shape)
    grads["bp"] += dpred.sum()
    dconcat = dpred[:, None] * wp[None, :]
    off = 0
    if model != "MLP":
        d = p["uG"].shape[1]
        dg = dconcat[:, :d]
        off = d
        np.add.at(grads["uG"], u, dg * p["iG"][item])
        np.add.at(grads["iG"], item, dg * p["uG"][u])
    if model != "GMF":
        dx = dconcat[:, off:]
        for l in range(num_layers, 0, -1):
            dz = dx * (cache["acts"][l - 1] > 0)
            grads[f"W{l}"] += dz.T @ cache["drops"][l - 1]
            grads[f"b{l}"] += dz.sum(0)
            dx = dz @ p[f"W{l}"]
            if masks is not None:
                dx = dx * masks[l - 1]
        dm = p["uM"].shape[1]
        np.add.at(grads["uM"], u, dx[:, :dm])
        np.add.at(grads["iM"], item, dx[:, dm:])


def neumf_grad(p, u, i, j, reg_1, reg_2, num_layers, loss_type=LOSS_BPR, model="NeuMF", gamma=1e-10,
               masks_pos=None, masks_neg=None, dtype=np.float64):
    """Loss and dense gradients of NeuMF.calc_loss (NeuMFRecommender.py:139-169); point-wise
    losses: `j` holds the labels."""
    p = {k: np.asarray(v, dtype) for k, v in p.items()}
    u = np.asarray(u, np.int64)
    i = np.asarray(i, np.int64)
    grads = {k: np.zeros_like(v) for k, v in p.items()}
    pointwise = loss_type in (LOSS_CL, LOSS_SL)
    pos, cpos = neumf_forward(p, u, i, num_layers, model, masks_pos, dtype)
    if pointwise:
        y = np.asarray(j, dtype)
        if loss_type == LOSS_CL:
            terms = np.maximum(pos, 0) - pos * y + np.log1p(np.exp(-np.abs(pos)))
            cp = _sigmoid(pos) - y
        else:
            terms = (pos - y) ** 2
            cp = 2.0 * (pos - y)
        loss = terms.sum(dtype=dtype)
        _backward_rows(p, cpos, cp, num_layers, model, masks_pos, grads)
    else:
        j = np.asarray(j, np.int64)
        neg, cneg = neumf_forward(p, u, j, num_layers, model, masks_neg, dtype)
        terms, cp, cn = pair_loss_coef(pos, neg, loss_type, dtype(gamma))
        loss = terms.sum(dtype=dtype)
        _backward_rows(p, cpos, cp, num_layers, model, masks_pos, grads)
        _backward_rows(p, cneg, cn, num_layers, model, masks_neg, grads)

    # ---- regularisers, term by term as NeuMFRecommender.py:149-167 writes them
    def l1(tab, idx, w=1.0):
        rows = p[tab][idx]
        np.add.at(grads[tab], idx, w * reg_1 * np.sign(rows))
        return w * reg_1 * np.abs(rows).sum(dtype=dtype)

    def fro(tab, idx, w=1.0):
        rows = p[tab][idx]
        n = np.sqrt((rows * rows).sum(dtype=dtype))
        if n > 0:
            np.add.at(grads[tab], idx, w * reg_2 * rows / n)
        return w * reg_2 * n

    loss += l1("iG", i) + l1("iM", i) + fro("iG", i) + fro("iM", i)
    if not pointwise:
        loss += l1("iG", j, 2.0) + fro("iG", j, 2.0)      # embed_item_GMF(neg_item) twice, MLP never
    loss += l1("uG", u) + l1("uM", u) + fro("uG", u) + fro("uM", u)
    return float(loss), grads


def neumf_rank(p, us, cands, topk, num_layers, model="NeuMF"):
    """NeuMFRecommender.py:171-209 (eval mode: no dropout), stable descending order."""
    us = np.asarray(us, np.int64)
    cands = np.asarray(cands, np.int64)
    B, C = cands.shape
    pred, _ = neumf_forward(p, np.repeat(us, C), cands.reshape(-1), num_layers, model, None, np.float32)
    scores = pred.reshape(B, C)
    order = np.argsort(-scores, axis=1, kind="stable")
    return np.take_along_axis(cands, order, axis=1)[:, :topk].astype(np.float32), scores


def neumf_full_rank(p, u, topk, num_layers, model="NeuMF"):
    """NeuMFRecommender.py:211-233."""
    I = np.asarray(p["iG"]).shape[0]
    pred, _ = neumf_forward(p, np.full(I, u, np.int64), np.arange(I), num_layers, model, None, np.float32)
    return np.argsort(-pred, kind="stable")[:topk].astype(np.int64)


# --------------------------------------------------------------------------------------------------
# Dropout masks of the HIP path (csrc/neumf.hip: drop_keep): a counter hash of (seed, layer, element),
# restated here so that training steps WITH dropout can be compared bit for bit in the mask.
# --------------------------------------------------------------------------------------------------
def drop_keep(seed, stream, idx, p):
    m32 = np.uint64(0xFFFFFFFF)
    thresh = np.uint64(min(int(np.float32(p).astype(np.float64) * 4294967296.0), 4294967295))
    idx = np.asarray(idx, np.uint64)
    seed = np.uint64(seed)
    h = _mix32((idx & m32) ^ (seed & m32))
    h = _mix32((h + (idx >> np.uint64(32)) * np.uint64(0x9E3779B9) + (seed >> np.uint64(32))
                + np.uint64(stream) * np.uint64(0x85EBCA6B)) & m32)
    return h >= thresh


def dropout_masks(seed, rows, d, num_layers, p, dtype=np.float64):
    """masks[l-1][r, c] for the input of Linear l (width 2*dm / 2^(l-1)), rows = global row ids of
    the step (positives 0..B-1, negatives B..2B-1), already scaled by 1/(1-p)."""
    rows = np.asarray(rows, np.uint64)
    w = 2 * d * (1 << (num_layers - 1))
    scale = dtype(np.float32(1.0) / (np.float32(1.0) - np.float32(p)))
    out = []
    for l in range(1, num_layers + 1):
        idx = rows[:, None] * np.uint64(w) + np.arange(w, dtype=np.uint64)[None, :]
        out.append(drop_keep(seed, l, idx, p).astype(dtype) * scale)
        w //= 2
    return out
