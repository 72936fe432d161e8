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
               masks_pos=None, masks_neg=None, dtype=np.float64, bf16_points=None):
    """Loss and dense gradients of NeuMF.calc_loss (NeuMFRecommender.py:139-169); point-wise
    losses: `j` holds the labels.  bf16_points='fact' | 'plain': the same step with the roundings to bf16 of the HIP path's
    bf16-storage mode (neumf_grad_bf16 below; full model, no dropout)."""
    if bf16_points is not None:
        assert model == "NeuMF" and masks_pos is None and masks_neg is None
        return neumf_grad_bf16(p, u, i, j, reg_1, reg_2, num_layers, loss_type, gamma, bf16_points)
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

    loss += _regularisers(p, grads, u, i, j, reg_1, reg_2, pointwise, dtype)
    return float(loss), grads


def _regularisers(p, grads, u, i, j, reg_1, reg_2, pointwise, dtype):
    """the regularisers, term by term as NeuMFRecommender.py:149-167 writes them: their value, gradients added in place"""
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

    loss = l1("iG", i) + l1("iM", i) + fro("iG", i) + fro("iM", i)
    if not pointwise:
        loss += l1("iG", j, 2.0) + fro("iG", j, 2.0)      # embed_item_GMF(neg_item) twice, MLP never
    loss += l1("uG", u) + l1("uM", u) + fro("uG", u) + fro("uM", u)
    return loss


# --------------------------------------------------------------------------------------------------
# The bf16-STORAGE mode of the HIP path (daisy_neumf_ctx_set_precision level 2; BASELINE configs[3] "MLP via MFMA bf16"),
# restated with a round-to-nearest-even to bf16 at exactly the points where csrc/neumf.hip / csrc/neumf_tower.hip store
# or feed bf16; everything else is the arithmetic of NeuMFRecommender.py:118-169 above.  Rounding points:
#   'fact'  (the first layer through the tables: dropout 0, fewer distinct table rows than rows in the step)
#           T_u = bf16(uM W1[:, :dm]^T), T_i = bf16(iM W1[:, dm:]^T)        fp32 products of the fp32 tables and weights
#           x1  = bf16(relu((T_u[user] + T_i[item]) + b1))                  fp32 adds
#   'plain' x0  = bf16([uM[user] | iM[item]]),  x1 = bf16(relu(x0 bf16(W1)^T + b1))
#   both    x_l = bf16(relu(x_{l-1} bf16(W_l)^T + b_l)), l >= 2             bf16 x bf16 products, fp32 accumulation
#           pred = <Wp[:d], uG[user] * iG[item]> + <Wp[d:], x_L> + bp        fp32
#           dZ_L = bf16(dpred Wp[d:] . [x_L > 0]);   gWp = sum dpred [g | x_L];   gbp = sum dpred
#           gW_l = dZ_l^T x_{l-1},  gb_l = sum_r dZ_l   (the stored bf16 values, fp32 accumulation)
#           dZ_{l-1} = bf16((dZ_l bf16(W_l)) . [x_{l-1} > 0])
#   'fact'  S_u = sum of dZ_1 over a user's rows (fp32), S_i likewise;  g uM = S_u W1[:, :dm],  gW1[:, :dm] = S_u^T uM  (fp32)
#   'plain' gW_1 = dZ_1^T x0,  dX0 = bf16(dZ_1 bf16(W1)) scattered to the tables
#   'inputs' (precision level 1: bf16 MFMA inputs, fp32 storage) every activation and gradient is STORED in fp32 and rounded
#           to bf16 only on its way into a product: x_l = relu(bf16(x_{l-1}) bf16(W_l)^T + b_l), gW_l = bf16(dZ_l)^T bf16(x_{l-1}),
#           dZ_{l-1} = (bf16(dZ_l) bf16(W_l)) . [x_{l-1} > 0]; bias and predict-layer gradients from the fp32 values
# The GMF branch and the regularisers never leave fp32.  Accumulations run in fp64 here (the device's fp32 accumulation
# differs from it by ~1e-6 relative, three orders below one bf16 rounding).
# --------------------------------------------------------------------------------------------------
def bf16_round(x):
    """float32 -> the nearest bf16 (ties to even), returned as float32 (csrc/neumf_internal.h: bf16_rne / bf16_pack2)"""
    x = np.ascontiguousarray(x, np.float32)
    u = x.view(np.uint32)
    r = ((u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) >> np.uint32(16)) << np.uint32(16)
    return r.view(np.float32).reshape(x.shape)


def neumf_grad_bf16(p, u, i, j, reg_1, reg_2, num_layers, loss_type=LOSS_BPR, gamma=1e-10, mode="fact"):
    """loss and dense gradients of one step in the bf16-storage mode (full model 'NeuMF', dropout 0); `mode` as above"""
    f32, f64 = np.float32, np.float64
    L = num_layers
    p32 = {k: np.asarray(v, f32) for k, v in p.items()}
    u, i = np.asarray(u, np.int64), np.asarray(i, np.int64)
    pointwise = loss_type in (LOSS_CL, LOSS_SL)
    users = u if pointwise else np.concatenate([u, u])
    items = i if pointwise else np.concatenate([i, np.asarray(j, np.int64)])
    B, d, dm = len(u), p32["uG"].shape[1], p32["uM"].shape[1]
    W16 = {l: bf16_round(p32[f"W{l}"]) for l in range(1, L + 1)}
    mm = lambda a, b: (a.astype(f64) @ b.astype(f64)).astype(f32)         # noqa: E731  (exact products, wide accumulation)
    W1 = p32["W1"]
    if mode == "inputs":
        return _neumf_grad_bf16_inputs(p32, u, i, j, users, items, reg_1, reg_2, L, loss_type, gamma, pointwise)
    if mode == "fact":
        Tu, Ti = bf16_round(mm(p32["uM"], W1[:, :dm].T)), bf16_round(mm(p32["iM"], W1[:, dm:].T))
        x = [None, bf16_round(np.maximum((Tu[users] + Ti[items]) + p32["b1"], f32(0)))]
    else:
        x0 = bf16_round(np.concatenate([p32["uM"][users], p32["iM"][items]], axis=1))
        x = [x0, bf16_round(np.maximum(mm(x0, W16[1].T) + p32["b1"], f32(0)))]
    for l in range(2, L + 1):
        x.append(bf16_round(np.maximum(mm(x[l - 1], W16[l].T) + p32[f"b{l}"], f32(0))))
    g = p32["uG"][users] * p32["iG"][items]
    wp = p32["Wp"].reshape(-1)
    concat = np.concatenate([g, x[L]], axis=1)
    pred = concat.astype(f64) @ wp.astype(f64) + f64(p32["bp"].reshape(-1)[0])
    if pointwise:
        y = np.asarray(j, f64)
        if loss_type == LOSS_CL:
            terms = np.maximum(pred, 0) - pred * y + np.log1p(np.exp(-np.abs(pred)))
            dpred = _sigmoid(pred) - y
        else:
            terms = (pred - y) ** 2
            dpred = 2.0 * (pred - y)
    else:
        terms, cp, cn = pair_loss_coef(pred[:B], pred[B:], loss_type, f64(gamma))
        dpred = np.concatenate([cp, cn])
    loss = terms.sum(dtype=f64)
    dp32 = dpred.astype(f32)
    grads = {k: np.zeros(v.shape, f64) for k, v in p32.items()}
    grads["Wp"] += (dp32[:, None].astype(f64) * concat).sum(0).reshape(p32["Wp"].shape)
    grads["bp"] += dp32.astype(f64).sum()
    np.add.at(grads["uG"], users, (dp32[:, None] * wp[None, :d]).astype(f64) * p32["iG"][items])
    np.add.at(grads["iG"], items, (dp32[:, None] * wp[None, :d]).astype(f64) * p32["uG"][users])
    dz = bf16_round(np.where(x[L] > 0, dp32[:, None] * wp[None, d:], f32(0)))
    for l in range(L, 1, -1):
        grads[f"W{l}"] += dz.astype(f64).T @ x[l - 1].astype(f64)
        grads[f"b{l}"] += dz.astype(f64).sum(0)
        dz = bf16_round(np.where(x[l - 1] > 0, mm(dz, W16[l]), f32(0)))
    grads["b1"] += dz.astype(f64).sum(0)
    if mode == "fact":
        Su, Si = np.zeros((p32["uM"].shape[0], dz.shape[1]), f64), np.zeros((p32["iM"].shape[0], dz.shape[1]), f64)
        np.add.at(Su, users, dz.astype(f64))
        np.add.at(Si, items, dz.astype(f64))
        grads["uM"] += Su @ W1[:, :dm].astype(f64)
        grads["iM"] += Si @ W1[:, dm:].astype(f64)
        grads["W1"][:, :dm] += Su.T @ p32["uM"].astype(f64)
        grads["W1"][:, dm:] += Si.T @ p32["iM"].astype(f64)
    else:
        grads["W1"] += dz.astype(f64).T @ x[0].astype(f64)
        dx0 = bf16_round(mm(dz, W16[1])).astype(f64)
        np.add.at(grads["uM"], users, dx0[:, :dm])
        np.add.at(grads["iM"], items, dx0[:, dm:])
    p64 = {k: v.astype(f64) for k, v in p32.items()}
    loss += _regularisers(p64, grads, u, i, None if pointwise else np.asarray(j, np.int64), reg_1, reg_2, pointwise, f64)
    return float(loss), grads


def _criterion(pred, B, j, loss_type, gamma, pointwise):
    f64 = np.float64
    if pointwise:
        y = np.asarray(j, f64)
        if loss_type == LOSS_CL:
            return (np.maximum(pred, 0) - pred * y + np.log1p(np.exp(-np.abs(pred)))).sum(dtype=f64), _sigmoid(pred) - y
        return ((pred - y) ** 2).sum(dtype=f64), 2.0 * (pred - y)
    terms, cp, cn = pair_loss_coef(pred[:B], pred[B:], loss_type, f64(gamma))
    return terms.sum(dtype=f64), np.concatenate([cp, cn])


def _neumf_grad_bf16_inputs(p32, u, i, j, users, items, reg_1, reg_2, L, loss_type, gamma, pointwise):
    """precision level 1 (see the table above): fp32 storage, operands rounded to bf16 on their way into every product"""
    f32, f64 = np.float32, np.float64
    B, d, dm = len(u), p32["uG"].shape[1], p32["uM"].shape[1]
    W16 = {l: bf16_round(p32[f"W{l}"]) for l in range(1, L + 1)}
    mm = lambda a, b: (a.astype(f64) @ b.astype(f64)).astype(f32)         # noqa: E731
    x = [np.concatenate([p32["uM"][users], p32["iM"][items]], axis=1)]
    for l in range(1, L + 1):
        x.append(np.maximum(mm(bf16_round(x[l - 1]), W16[l].T) + p32[f"b{l}"], f32(0)))
    g = p32["uG"][users] * p32["iG"][items]
    wp = p32["Wp"].reshape(-1)
    concat = np.concatenate([g, x[L]], axis=1)
    pred = concat.astype(f64) @ wp.astype(f64) + f64(p32["bp"].reshape(-1)[0])
    loss, dpred = _criterion(pred, B, j, loss_type, gamma, pointwise)
    dp32 = dpred.astype(f32)
    grads = {k: np.zeros(v.shape, f64) for k, v in p32.items()}
    grads["Wp"] += (dp32[:, None].astype(f64) * concat).sum(0).reshape(p32["Wp"].shape)
    grads["bp"] += dp32.astype(f64).sum()
    np.add.at(grads["uG"], users, (dp32[:, None] * wp[None, :d]).astype(f64) * p32["iG"][items])
    np.add.at(grads["iG"], items, (dp32[:, None] * wp[None, :d]).astype(f64) * p32["uG"][users])
    dz = np.where(x[L] > 0, dp32[:, None] * wp[None, d:], f32(0)).astype(f32)
    for l in range(L, 0, -1):
        dz16 = bf16_round(dz)
        grads[f"W{l}"] += dz16.astype(f64).T @ bf16_round(x[l - 1]).astype(f64)
        grads[f"b{l}"] += dz.astype(f64).sum(0)
        dx = mm(dz16, W16[l])
        dz = np.where(x[l - 1] > 0, dx, f32(0)).astype(f32) if l > 1 else dx
    np.add.at(grads["uM"], users, dz[:, :dm].astype(f64))
    np.add.at(grads["iM"], items, dz[:, dm:].astype(f64))
    p64 = {k: v.astype(f64) for k, v in p32.items()}
    loss += _regularisers(p64, grads, u, i, None if pointwise else np.asarray(j, np.int64), reg_1, reg_2, pointwise, f64)
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
