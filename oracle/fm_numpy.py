"""CPU restatement of the reference's FM recommender — TEST INFRASTRUCTURE ONLY.

`daisy/model/FMRecommender.py` is MF plus three bias terms:

    pred(u, item) = <P[u], Q[item]> + u_bias[u] + i_bias[item] + bias_        (FMRecommender.py:61-68)

with the SAME loss construction as MF (criterion on (pos, neg) or (pos, label), non-squared L1 /
Frobenius regularisers on the gathered embedding rows only — the biases are not regularised:
FMRecommender.py:70-93).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may
import this module; the product path never does.

Pinned by tests/golden/kat_fm.npz, generated from the real reference by
tests/golden/make_golden_fm.py.
"""
import numpy as np

from .bpr_mf_numpy import (LOSS_CL, LOSS_SL, LOSS_BPR, DenseAdam, _sigmoid,  # noqa: F401
                           pair_loss_coef)


def fm_forward(P, Q, bu, bi, b0, u, i, dtype=np.float32):
    """FMRecommender.py:61-68 in the reference's order of additions: dot, then += (bu + bi) + b0."""
    P, Q = np.asarray(P, dtype), np.asarray(Q, dtype)
    bu, bi = np.asarray(bu, dtype).reshape(-1), np.asarray(bi, dtype).reshape(-1)
    b0 = dtype(np.asarray(b0).reshape(-1)[0])
    return np.einsum("bk,bk->b", P[u], Q[i]) + ((bu[u] + bi[i]) + b0)


def fm_grad(P, Q, bu, bi, b0, u, i, j, reg_1, reg_2, loss_type=LOSS_BPR, gamma=1e-10,
            dtype=np.float64):
    """Loss and the dense gradients (gP, gQ, g_bu, g_bi, g_b0) autograd produces for one batch of
    FM.calc_loss (FMRecommender.py:70-93).  Point-wise losses: `j` holds the labels."""
    P, Q = np.asarray(P, dtype), np.asarray(Q, dtype)
    bu, bi = np.asarray(bu, dtype).reshape(-1), np.asarray(bi, dtype).reshape(-1)
    b0 = dtype(np.asarray(b0).reshape(-1)[0])
    u = np.asarray(u, np.int64)
    i = np.asarray(i, np.int64)
    pointwise = loss_type in (LOSS_CL, LOSS_SL)
    pu, qi = P[u], Q[i]
    pos = np.einsum("bk,bk->b", pu, qi) + bu[u] + bi[i] + b0
    gP, gQ = np.zeros_like(P), np.zeros_like(Q)
    g_bu, g_bi = np.zeros_like(bu), np.zeros_like(bi)

    def _fro(x, n):
        return x / n if n > 0 else np.zeros_like(x)

    nU = np.sqrt((pu * pu).sum(dtype=dtype))
    nI = np.sqrt((qi * qi).sum(dtype=dtype))
    loss = reg_1 * np.abs(qi).sum(dtype=dtype) + reg_2 * nI + reg_1 * np.abs(pu).sum(dtype=dtype) + reg_2 * nU
    if pointwise:
        y = np.asarray(j, dtype)
        if loss_type == LOSS_CL:
            terms = np.maximum(pos, 0) - pos * y + np.log1p(np.exp(-np.abs(pos)))
            cp = _sigmoid(pos) - y
        else:
            terms = (pos - y) ** 2
            cp = 2.0 * (pos - y)
        cn = np.zeros_like(cp)
        loss += terms.sum(dtype=dtype)
        np.add.at(gP, u, cp[:, None] * qi + reg_1 * np.sign(pu) + reg_2 * _fro(pu, nU))
    else:
        j = np.asarray(j, np.int64)
        qj = Q[j]
        neg = np.einsum("bk,bk->b", pu, qj) + bu[u] + bi[j] + b0
        terms, cp, cn = pair_loss_coef(pos, neg, loss_type, dtype(gamma))
        nJ = np.sqrt((qj * qj).sum(dtype=dtype))
        loss += terms.sum(dtype=dtype) + reg_1 * np.abs(qj).sum(dtype=dtype) + reg_2 * nJ
        np.add.at(gP, u, cp[:, None] * qi + cn[:, None] * qj + reg_1 * np.sign(pu) + reg_2 * _fro(pu, nU))
        np.add.at(gQ, j, cn[:, None] * pu + reg_1 * np.sign(qj) + reg_2 * _fro(qj, nJ))
        np.add.at(g_bi, j, cn)
    np.add.at(gQ, i, cp[:, None] * pu + reg_1 * np.sign(qi) + reg_2 * _fro(qi, nI))
    np.add.at(g_bi, i, cp)
    np.add.at(g_bu, u, cp + cn)
    g_b0 = (cp + cn).sum(dtype=dtype)
    return float(loss), gP, gQ, g_bu, g_bi, g_b0


def fm_sgd_step(P, Q, bu, bi, b0, u, i, j, lr, reg_1, reg_2, loss_type=LOSS_BPR, gamma=1e-10,
                dtype=np.float64):
    """zero_grad / calc_loss / backward / SGD.step (AbstractRecommender.py:119-126) for FM."""
    loss, gP, gQ, g_bu, g_bi, g_b0 = fm_grad(P, Q, bu, bi, b0, u, i, j, reg_1, reg_2, loss_type, gamma, dtype)
    f = np.float32
    lr = dtype(lr)
    return (loss, (np.asarray(P, dtype) - lr * gP).astype(f), (np.asarray(Q, dtype) - lr * gQ).astype(f),
            (np.asarray(bu, dtype).reshape(-1) - lr * g_bu).astype(f),
            (np.asarray(bi, dtype).reshape(-1) - lr * g_bi).astype(f),
            f(dtype(np.asarray(b0).reshape(-1)[0]) - lr * g_b0))


def fm_rank(P, Q, bu, bi, b0, us, cands, topk):
    """FMRecommender.py:105-123 (fp32, the reference's order of additions, stable descending sort)."""
    f = np.float32
    P, Q = np.asarray(P, f), np.asarray(Q, f)
    bu, bi = np.asarray(bu, f).reshape(-1), np.asarray(bi, f).reshape(-1)
    b0 = f(np.asarray(b0).reshape(-1)[0])
    us, cands = np.asarray(us, np.int64), np.asarray(cands, np.int64)
    scores = np.einsum("bk,bck->bc", P[us], Q[cands]).astype(f)
    scores = scores + ((bu[us][:, None] + bi[cands]) + b0)
    order = np.argsort(-scores, axis=1, kind="stable")
    return np.take_along_axis(cands, order, axis=1)[:, :topk].astype(np.float32), scores


def fm_full_rank(P, Q, bu, bi, b0, u, topk):
    """FMRecommender.py:125-133."""
    f = np.float32
    scores = (np.asarray(Q, f) @ np.asarray(P, f)[u]).astype(f)
    scores = scores + ((f(np.asarray(bu, f).reshape(-1)[u]) + np.asarray(bi, f).reshape(-1))
                       + f(np.asarray(b0).reshape(-1)[0]))
    return np.argsort(-scores, kind="stable")[:topk].astype(np.int64)
