"""CPU restatement of the reference's LightGCN recommender — TEST INFRASTRUCTURE ONLY.

Follows `daisy/model/LightGCNRecommender.py`:
  get_norm_adj_mat (:74-107)  A = [[0, R], [R^T, 0]] over N = U + I nodes, entries 1 (duplicate
                              interactions collapse), deg = (#distinct neighbours) + 1e-7,
                              A_hat = D^-1/2 A D^-1/2 evaluated in float64 and stored as float32
  forward          (:117-129) E_0 = [embed_user; embed_item];  E_{k+1} = A_hat E_k;
                              out = mean(E_0 .. E_L)  (the propagation is recomputed for EVERY batch)
  calc_loss        (:131-169) criterion on <out_u, out_i>, <out_u, out_j>; non-squared L1 / Frobenius
                              regularisers on the EGO (layer-0) rows of the batch
  predict / rank / full_rank (:171-210) on the propagated embeddings
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this module.
Pinned by tests/golden/kat_lightgcn.npz (tests/golden/make_golden_lightgcn.py).
"""
import numpy as np

from .bpr_mf_numpy import LOSS_BPR, LOSS_CL, LOSS_SL, _sigmoid, pair_loss_coef  # noqa: F401


def norm_adj_csr(users, items, user_num, item_num):
    """CSR (indptr int64[N+1], col int32[nnz], val float32[nnz]) of A_hat, rows and columns ascending
    (LightGCNRecommender.py:74-107)."""
    users = np.asarray(users, np.int64)
    items = np.asarray(items, np.int64)
    N = user_num + item_num
    pairs = np.unique(users * item_num + items)                  # duplicate interactions collapse (:88-90)
    u, i = pairs // item_num, pairs % item_num
    rows = np.concatenate([u, i + user_num])
    cols = np.concatenate([i + user_num, u])
    order = np.lexsort((cols, rows))
    rows, cols = rows[order], cols[order]
    deg = np.bincount(rows, minlength=N).astype(np.float64) + 1e-7    # (A > 0).sum(1) + 1e-7  (:93-95)
    dinv = np.power(deg, -0.5)
    val = ((dinv[rows] * 1.0) * dinv[cols]).astype(np.float32)        # D * A * D, then FloatTensor (:97-105)
    indptr = np.zeros(N + 1, np.int64)
    np.cumsum(np.bincount(rows, minlength=N), out=indptr[1:])
    return indptr, cols.astype(np.int32), val


def spmm(graph, X):
    indptr, col, val = graph
    X = np.asarray(X)
    out = np.zeros_like(X)
    rows = np.repeat(np.arange(len(indptr) - 1), np.diff(indptr))
    np.add.at(out, rows, val[:, None].astype(X.dtype) * X[col])
    return out


def propagate(graph, E0, num_layers):
    """LightGCNRecommender.py:117-129: mean of E_0 .. E_L."""
    acc, E = E0.copy(), E0
    for _ in range(num_layers):
        E = spmm(graph, E)
        acc = acc + E
    return acc / (num_layers + 1)


def lightgcn_grad(graph, P, Q, u, i, j, reg_1, reg_2, num_layers, loss_type=LOSS_BPR, gamma=1e-10,
                  dtype=np.float64):
    """Loss and dense gradients (gP, gQ) of LightGCN.calc_loss; point-wise losses: j = labels."""
    P, Q = np.asarray(P, dtype), np.asarray(Q, dtype)
    U = P.shape[0]
    u = np.asarray(u, np.int64)
    i = np.asarray(i, np.int64)
    E0 = np.concatenate([P, Q], 0)
    out = propagate(graph, E0, num_layers)
    ou, oi = out[u], out[U + i]
    pos = np.einsum("bk,bk->b", ou, oi)
    G = np.zeros_like(E0)
    pointwise = loss_type in (LOSS_CL, LOSS_SL)
    if pointwise:
        y = np.asarray(j, dtype)
        if loss_type == LOSS_CL:
            terms = np.maximum(pos, 0) - pos * y + np.log1p(np.exp(-np.abs(pos)))
            cp = _sigmoid(pos) - y
        else:
            terms = (pos - y) ** 2
            cp = 2.0 * (pos - y)
        np.add.at(G, u, cp[:, None] * oi)
    else:
        j = np.asarray(j, np.int64)
        oj = out[U + j]
        neg = np.einsum("bk,bk->b", ou, oj)
        terms, cp, cn = pair_loss_coef(pos, neg, loss_type, dtype(gamma))
        np.add.at(G, u, cp[:, None] * oi + cn[:, None] * oj)
        np.add.at(G, U + j, cn[:, None] * ou)
    np.add.at(G, U + i, cp[:, None] * ou)
    loss = terms.sum(dtype=dtype)
    # d out / d E0: out = 1/(L+1) sum_k A^k E0 with A symmetric  ->  dE0 = 1/(L+1) sum_k A^k G (Horner)
    T = G.copy()
    for _ in range(num_layers):
        T = G + spmm(graph, T)
    dE0 = T / (num_layers + 1)

    def reg(rows_idx, table_off):
        rows = E0[table_off + rows_idx]
        n = np.sqrt((rows * rows).sum(dtype=dtype))
        np.add.at(dE0, table_off + rows_idx, reg_1 * np.sign(rows) + (reg_2 * rows / n if n > 0 else 0.0))
        return reg_1 * np.abs(rows).sum(dtype=dtype) + reg_2 * n

    loss += reg(u, 0) + reg(i, U)
    if not pointwise:
        loss += reg(j, U)
    return float(loss), dE0[:U], dE0[U:]


def lightgcn_rank(graph, P, Q, us, cands, topk, num_layers):
    """LightGCNRecommender.py:178-200 (fp32, stable descending order)."""
    P, Q = np.asarray(P, np.float32), np.asarray(Q, np.float32)
    U = P.shape[0]
    out = propagate(graph, np.concatenate([P, Q], 0), num_layers)
    us, cands = np.asarray(us, np.int64), np.asarray(cands, np.int64)
    scores = np.einsum("bk,bck->bc", out[us], out[U + cands])
    order = np.argsort(-scores, axis=1, kind="stable")
    return np.take_along_axis(cands, order, axis=1)[:, :topk].astype(np.float32), scores


def lightgcn_full_rank(graph, P, Q, u, topk, num_layers):
    P, Q = np.asarray(P, np.float32), np.asarray(Q, np.float32)
    U = P.shape[0]
    out = propagate(graph, np.concatenate([P, Q], 0), num_layers)
    scores = out[U:] @ out[u]
    return np.argsort(-scores, kind="stable")[:topk].astype(np.int64)
