"""Device-side replacements for the two host loops of daisy/utils/utils.py that sit either side
of the MF hot path at scale:

* ``get_ur``               utils.py:19-34  (Python iterrows loop)       -> ``user_csr`` (device CSR), and ``get_ur`` /
                                                                          ``get_ir`` with the reference's return type
                                                                          (dict of sets) built without the row loop
* ``build_candidates_set`` utils.py:53-85  (O(U*I) np.setdiff1d loop)   -> same name, same return
"""
from __future__ import annotations

from collections import defaultdict

import numpy as np
import torch

from .. import ops


def _grouped_sets(keys, vals):
    out = defaultdict(set)
    keys, vals = np.asarray(keys).astype(np.int64), np.asarray(vals).astype(np.int64)
    if keys.size == 0:
        return out
    order = np.argsort(keys, kind="stable")
    ks, vs = keys[order], vals[order]
    cuts = np.flatnonzero(ks[1:] != ks[:-1]) + 1
    for k, chunk in zip(ks[np.r_[0, cuts]].tolist(), np.split(vs, cuts)):
        out[k] = set(chunk.tolist())
    return out


def get_ur(df):
    """utils.py:19-34: {user: set(items)} as a defaultdict(set) with int keys - the same object the reference's
    row loop builds (1.9 s per 100 k rows there), from one sort (host numpy: this dict is host data by contract)."""
    return _grouped_sets(df["user"].to_numpy(), df["item"].to_numpy())


def get_ir(df):
    """utils.py:36-51: {item: set(users)}"""
    return _grouped_sets(df["item"].to_numpy(), df["user"].to_numpy())


def user_csr(df_or_pairs, user_num, device="cuda"):
    """(indptr int64[U+1], sorted items int32[n]) of the (user, item) pairs of a DataFrame
    (columns 'user', 'item') or of a (users, items) pair of arrays."""
    if hasattr(df_or_pairs, "columns"):
        users, items = df_or_pairs["user"].to_numpy(), df_or_pairs["item"].to_numpy()
    else:
        users, items = df_or_pairs
    users = torch.as_tensor(np.asarray(users).astype(np.int32)).to(device)
    items = torch.as_tensor(np.asarray(items).astype(np.int32)).to(device)
    return ops.build_user_csr(users, items, user_num)


def _ur_pairs(ur):
    us = np.fromiter((u for u, s in ur.items() for _ in s), dtype=np.int32)
    it = np.fromiter((i for _, s in ur.items() for i in s), dtype=np.int32)
    return us, it


def build_candidates_set(test_ur, train_ur, config, drop_past_inter=True):
    """Same signature and return value as utils.py:53-85: (test_u, test_ucands) with
    test_ucands = [[u, np.ndarray(cand_num)], ...] in test_ur's key order.  Negatives come from the
    device generator (Philox keyed by config['seed']); truths are appended in ascending order."""
    if not drop_past_inter:
        raise NotImplementedError("drop_past_inter=False is not used by the drivers (test.py:112, tune.py:204)")
    if not torch.cuda.is_available():
        raise RuntimeError("daisyrec_amd.build_candidates_set needs a HIP device (no CPU fallback)")
    user_num, item_num, cand_num = config["user_num"], config["item_num"], config["cand_num"]
    dev = config.get("device", "cuda")
    ip_te, it_te = user_csr(_ur_pairs(test_ur), user_num, dev)
    ip_tr, it_tr = user_csr(_ur_pairs(train_ur), user_num, dev)
    test_u = list(test_ur.keys())
    users = torch.as_tensor(np.asarray(test_u, dtype=np.int64)).to(dev)
    cands = ops.build_candidates(ip_te, it_te, ip_tr, it_tr, users, item_num, cand_num,
                                 int(config.get("seed", 2022))).cpu().numpy()
    return test_u, [[u, cands[k]] for k, u in enumerate(test_u)]
