"""CPU restatement of the reference's Item2Vec recommender — TEST INFRASTRUCTURE ONLY.

Follows `daisy/model/Item2VecRecommender.py`:
  forward / calc_loss (:47-69)  pred = <S[target], S[context]> on ONE shared item table S;
                                loss = BCEWithLogitsLoss(sum)(pred, label)  ('CL', :38-39), no regulariser
  fit (:53-59)                  after training: user_embedding[u] = sum_{i in train_ur[u]} S[i]
  predict / rank / full_rank (:71-112)  MF scoring with (user_embedding, shared_embedding)
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this module.
Pinned by tests/golden/kat_item2vec.npz (tests/golden/make_golden_item2vec.py).
"""
import numpy as np

from .bpr_mf_numpy import _sigmoid, mf_full_rank, mf_rank  # noqa: F401


def item2vec_grad(S, target, context, label, dtype=np.float64):
    """Loss and the dense gradient of the shared table for one batch (:47-69)."""
    S = np.asarray(S, dtype)
    t = np.asarray(target, np.int64)
    c = np.asarray(context, np.int64)
    y = np.asarray(label, dtype)
    st, sc = S[t], S[c]
    x = np.einsum("bk,bk->b", st, sc)
    terms = np.maximum(x, 0) - x * y + np.log1p(np.exp(-np.abs(x)))
    coef = _sigmoid(x) - y
    g = np.zeros_like(S)
    np.add.at(g, t, coef[:, None] * sc)
    np.add.at(g, c, coef[:, None] * st)
    return float(terms.sum(dtype=dtype)), g


def build_user_embedding(S, train_ur, user_num):
    """:56-59  user_embedding[u] = S[list(train_ur[u])].sum(0) (fp32 like the reference; users absent
    from train_ur keep their initial rows - the caller passes those in)."""
    S = np.asarray(S, np.float32)
    out = {}
    for u, items in train_ur.items():
        out[int(u)] = S[np.asarray(sorted(items), np.int64)].sum(0, dtype=np.float32)
    return out
