"""Tensor-level operators over the C ABI (``include/daisyrec_amd.h``).

PyTorch is plumbing here: it owns device memory and streams; every operator
hands raw device pointers + the current HIP stream to ``libdaisyrec_hip.so``.
Nothing in this module computes on the host or falls back to torch ops.
"""
from __future__ import annotations

import ctypes as C
import weakref

import torch

from . import _native as N
from ._native import check, lib

LOSS_IDS = {"BPR": N.LOSS_BPR, "HL": N.LOSS_HL, "TL": N.LOSS_TL, "CL": N.LOSS_CL, "SL": N.LOSS_SL}
POINTWISE_LOSSES = (N.LOSS_CL, N.LOSS_SL)
ITEM_MODES = {"atomic": N.ITEM_ATOMIC, "sorted": N.ITEM_SORTED, "chunked": N.ITEM_CHUNKED,
              "fused": N.ITEM_FUSED}
SMALL_BATCH_MAX = 256      # kSmallBatchMax (csrc/bpr_internal.h): fit_epoch_sgd runs such epochs in one persistent workgroup
ORDER_MODES = {"identity": N.ORDER_IDENTITY, "perm": N.ORDER_PERM, "feistel": N.ORDER_FEISTEL}


def loss_id(loss_type: str) -> int:
    key = str(loss_type).upper()
    if key not in LOSS_IDS:
        # MFRecommender.py:90-91 / AbstractRecommender.py:91
        raise NotImplementedError(f"Invalid loss type: {loss_type}")
    return LOSS_IDS[key]


def _ptr(t, dtype, name):
    if t is None:
        return None
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a torch.Tensor, got {type(t).__name__}")
    if not t.is_cuda:
        raise RuntimeError(f"{name}: tensor is on {t.device}; the HIP path needs device memory "
                           "(there is no CPU fallback)")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: tensor must be contiguous")
    return C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ws(nbytes, device):
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


class TrainIndex:
    """Owner of a native ``daisy_train_index``: the training triples in CSR order + their item entries
    sorted by item, built once per fit (BasicDataset's triple array is immutable, dataset.py:21).
    Raises ValueError when an id lies outside the tables (the reference: IndexError in nn.Embedding)."""

    def __init__(self, triples, user_num: int, item_num: int, user_base: int = 0, user_sorted=None, pointwise=False):
        """pointwise: rows are (user, item, label) (CL / SL, sampler.py:93-98): one item entry per row"""
        if user_sorted is None:
            user_sorted = triples_user_sorted(triples)
        self.triples = triples            # kept alive: a user-sorted array is indexed in place
        self.n = int(triples.shape[0])
        self.user_num, self.item_num = int(user_num), int(item_num)
        self._h = C.c_void_p()
        with torch.cuda.device(triples.device):
            check(lib.daisy_train_index_create(C.byref(self._h), _ptr(triples, torch.int32, "triples"), self.n,
                                               self.user_num, self.item_num, int(user_base),
                                               (N.PLAN_TRIPLES_USER_SORTED if user_sorted else 0)
                                               | (N.PLAN_POINTWISE if pointwise else 0), _stream()))

    @property
    def nbytes(self):
        return int(lib.daisy_train_index_bytes(self._h))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            check(lib.daisy_train_index_destroy(self._h))
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class EpochPlan:
    """Owner of a native ``daisy_epoch_plan``: one epoch laid out batch by batch in HBM
    (replaces a pass over DataLoader(BasicDataset(triples)), dataset.py:5-27)."""

    def __init__(self, max_triples: int, user_num: int, item_num: int, device="cuda"):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("EpochPlan needs a HIP device (no CPU fallback)")
        self.max_triples = int(max_triples)
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            check(lib.daisy_epoch_plan_create(C.byref(self._h), self.max_triples, int(user_num),
                                              int(item_num)))

    def build(self, triples, batch_size, order="identity", perm=None, seed=0, epoch=0, user_base=0,
              n_triples=None, user_sorted=False, pointwise=False, validate=None):
        """user_sorted=True promises that `triples` is sorted by user (use `triples_user_sorted`
        to check once): grouping by user then costs one radix pass instead of a full sort.
        validate: check the id ranges after the build (one host sync; ValueError like the reference's
        IndexError).  Default: the first build over a given triple array (and every explicit permutation)."""
        n = triples.shape[0] if n_triples is None else int(n_triples)
        mode = ORDER_MODES[order] if isinstance(order, str) else int(order)
        flags = (N.PLAN_TRIPLES_USER_SORTED if user_sorted else 0) | (N.PLAN_POINTWISE if pointwise else 0)
        check(lib.daisy_epoch_plan_build(self._h, _ptr(triples, torch.int32, "triples"), n,
                                         _ptr(perm, torch.int64, "perm"), mode, int(seed), int(epoch),
                                         int(batch_size), int(user_base), flags, _stream()))
        if validate is None:       # the same tensor object, unchanged since its last validated build, is not re-read
            ref = getattr(self, "_validated", None)
            same = ref is not None and ref[0]() is triples and ref[1:] == (n, int(user_base), triples._version)
            validate = (not same) or mode == N.ORDER_PERM
        if validate:
            check(lib.daisy_epoch_plan_validate(self._h, _stream()))
            self._validated = (weakref.ref(triples), n, int(user_base), triples._version)
        return self

    def build_indexed(self, index: "TrainIndex", batch_size, order="identity", perm=None, seed=0, epoch=0):
        """The partitioned layout (32 B per interaction, two one-digit partitions of the static index
        instead of two radix sorts); feeds the staged step (item_mode 'fused') only."""
        mode = ORDER_MODES[order] if isinstance(order, str) else int(order)
        check(lib.daisy_epoch_plan_build_indexed(self._h, index._h, _ptr(perm, torch.int64, "perm"), mode,
                                                 int(seed), int(epoch), int(batch_size), _stream()))
        return self

    def build_positions(self, index: "TrainIndex", positions, n_total, batch_size):
        """One rank's share of an epoch over n_total rows: `index` holds this rank's rows, positions[r] (int64,
        distinct, in [0, n_total)) the place of its row r in the epoch order; batch k = the held rows with position
        in [k*B, (k+1)*B) - the union over the ranks is batch k of the single-device epoch.  Batches differ in
        size (`batch_rows`); one host sync."""
        check(lib.daisy_epoch_plan_build_positions(self._h, index._h, _ptr(positions, torch.int64, "positions"),
                                                   int(n_total), int(batch_size), _stream()))
        return self

    def batch_rows(self, k):
        """rows of batch k held by this plan (host value)"""
        return int(lib.daisy_epoch_plan_batch_rows(self._h, int(k)))

    @property
    def num_batches(self):
        return int(lib.daisy_epoch_plan_num_batches(self._h))

    def read_batch(self, k, batch_size):
        """(u, i, j, ent_item, ent_s, ent_u) of batch k as device tensors (inspection / tests)."""
        dev = self.device
        u, i, j = (torch.empty(batch_size, dtype=torch.int32, device=dev) for _ in range(3))
        ei, es, eu = (torch.empty(2 * batch_size, dtype=torch.int32, device=dev) for _ in range(3))
        B = C.c_int64(0)
        check(lib.daisy_epoch_plan_read_batch(self._h, int(k), _ptr(u, torch.int32, "u"),
                                              _ptr(i, torch.int32, "i"), _ptr(j, torch.int32, "j"),
                                              _ptr(ei, torch.int32, "ent_item"), _ptr(es, torch.int32, "ent_s"),
                                              _ptr(eu, torch.int32, "ent_u"), C.byref(B), _stream()))
        b = B.value
        return u[:b], i[:b], j[:b], ei[:2 * b], es[:2 * b], eu[:2 * b]

    @property
    def nbytes(self):
        return int(lib.daisy_epoch_plan_bytes(self._h))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            check(lib.daisy_epoch_plan_destroy(self._h))
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def triples_user_sorted(triples) -> bool:
    """One-off check (plumbing, not the hot path) that a device triple array is in CSR order."""
    u = triples[:, 0]
    return bool((u[1:] >= u[:-1]).all().item()) if u.numel() > 1 else True


def feistel_positions(n, seed, epoch=0, device="cuda"):
    """pos[t] = position of triple t in the device shuffle of (seed, epoch)."""
    out = torch.empty(n, dtype=torch.int64, device=device)
    check(lib.daisy_feistel_positions(int(n), int(seed), int(epoch), _ptr(out, torch.int64, "out"),
                                      _stream()))
    return out


def feistel_positions_at(ids, n, seed, epoch=0):
    """pos[k] = position of triple ids[k] in the device shuffle of all n triples (a rank's rows of a multi-GPU fit)."""
    out = torch.empty(ids.shape[0], dtype=torch.int64, device=ids.device)
    if ids.numel() == 0:            # a rank without rows (the native entry rejects n_ids == 0)
        return out
    check(lib.daisy_feistel_positions_at(_ptr(ids, torch.int64, "ids"), ids.shape[0], int(n), int(seed), int(epoch),
                                         _ptr(out, torch.int64, "out"), _stream()))
    return out


class BprContext:
    """Owner of a native ``daisy_bpr_ctx`` (per-step scratch on one GPU)."""

    def __init__(self, max_batch: int, d: int, user_num: int, item_num: int, device="cuda"):
        self.max_batch, self.d, self.user_num, self.item_num = int(max_batch), int(d), int(user_num), int(item_num)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("BprContext needs a HIP device (no CPU fallback)")
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            check(lib.daisy_bpr_ctx_create(C.byref(self._h), self.max_batch, self.d, self.user_num,
                                           self.item_num))
        # caller-owned (so torch.distributed can all-reduce them)
        self.stats = torch.zeros(N.STATS_LEN, dtype=torch.float64, device=self.device)
        self.epoch_acc = torch.zeros(2, dtype=torch.float64, device=self.device)
        self.gQ = torch.zeros(self.item_num, self.d, dtype=torch.float32, device=self.device)
        self._p_key = None        # (data_ptr, torch version counter) of the P the row-norm cache describes
        self._bias = None

    def _sync_norm_cache(self, P):
        """The staged step keeps |P[u]|^2 per row inside the context.  The native side tracks every write
        that goes through it; torch-side in-place edits of P show up in the tensor's version counter."""
        key = (P.data_ptr(), P._version)
        if key != self._p_key:
            check(lib.daisy_bpr_ctx_invalidate_cache(self._h))
            self._p_key = key

    def invalidate_cache(self):
        check(lib.daisy_bpr_ctx_invalidate_cache(self._h))
        self._p_key = None

    def set_p_stream(self, mode="auto"):
        """staged step: user rows read / written past the caches ('auto': by table size - beyond 512 MB; True / False)"""
        m = -1 if mode in ("auto", None, -1) else (1 if mode else 0)
        check(lib.daisy_bpr_ctx_set_p_stream(self._h, m))

    # -- the staged step in phases (multi-GPU form) ------------------------------------------------
    def staged_prenorm(self, P):
        self._sync_norm_cache(P)
        check(lib.daisy_bpr_staged_prenorm(self._h, _ptr(P, torch.float32, "P"),
                                           _ptr(self.stats, torch.float64, "stats"), _stream()))

    def staged_user(self, P, Q, lr, reg_1, reg_2, loss_type=N.LOSS_BPR, gamma=1e-10):
        check(lib.daisy_bpr_staged_user(self._h, _ptr(P, torch.float32, "P"), _ptr(Q, torch.float32, "Q"),
                                        int(loss_type), float(gamma), float(lr), float(reg_1), float(reg_2),
                                        _ptr(self.stats, torch.float64, "stats"), _stream()))

    def staged_item(self, lr, reg_1, reg_2, Q=None, gQ=None, cnt=None, loss_type=N.LOSS_BPR):
        """Q given: SGD on the touched item rows in place.  gQ + cnt given instead: the data term of the
        item gradient and the per-item (n_pos, n_neg) for a reduce-scatter."""
        check(lib.daisy_bpr_staged_item(self._h, int(loss_type), _ptr(Q, torch.float32, "Q"),
                                        _ptr(gQ, torch.float32, "gQ"), _ptr(cnt, torch.float32, "cnt"),
                                        float(lr), float(reg_1), float(reg_2),
                                        _ptr(self.stats, torch.float64, "stats"), _stream()))

    def staged_adam_catchup_users(self, P, adam):
        """multi-GPU Adam: the rows of P the current batch references -> step adam.t - 1 (zero-gradient replays)"""
        self._sync_norm_cache(P)
        f = torch.float32
        check(lib.daisy_bpr_staged_adam_catchup_users(
            self._h, _ptr(P, f, "P"), _ptr(adam.mP, f, "mP"), _ptr(adam.vP, f, "vP"), _ptr(adam.lastP, torch.int32, "lastP"),
            _ptr(adam.table, f, "table"), adam.BETA1, adam.BETA2, adam.EPS, adam.t, _stream()))

    def staged_user_adam(self, P, Q, adam, reg_1, reg_2, loss_type=N.LOSS_BPR, gamma=1e-10):
        f = torch.float32
        check(lib.daisy_bpr_staged_user_adam(
            self._h, _ptr(P, f, "P"), _ptr(Q, f, "Q"), int(loss_type), float(gamma), adam.lr, float(reg_1), float(reg_2),
            _ptr(adam.mP, f, "mP"), _ptr(adam.vP, f, "vP"), _ptr(adam.lastP, torch.int32, "lastP"), adam.BETA1, adam.BETA2,
            adam.EPS, adam.t, _ptr(self.stats, torch.float64, "stats"), _stream()))

    def item_apply_counts_adam(self, Q_rows, g_rows, cnt_rows, m_rows, v_rows, adam, reg_1, reg_2):
        """The row owner's DENSE Adam step over its block of Q after the reduce-scatter of (g, cnt); clears g and cnt."""
        f = torch.float32
        check(lib.daisy_item_apply_counts_adam(
            _ptr(Q_rows, f, "Q"), _ptr(g_rows, f, "g"), _ptr(cnt_rows, f, "cnt"), _ptr(m_rows, f, "m"), _ptr(v_rows, f, "v"),
            Q_rows.shape[0], Q_rows.shape[1], adam.lr, float(reg_1), float(reg_2), adam.BETA1, adam.BETA2, adam.EPS, adam.t,
            _ptr(self.stats, torch.float64, "stats"), _stream()))

    def staged_item_slices(self, item_bounds):
        """Cut the item pass of the current batch at the given items (host ints, bounds[0] = 0, bounds[-1] >= item_num,
        at most 16 slices): `staged_item_slice(s, ...)` then reduces the entries of items [bounds[s], bounds[s+1])."""
        b = (C.c_int32 * len(item_bounds))(*[int(x) for x in item_bounds])
        check(lib.daisy_bpr_staged_item_slices(self._h, b, len(item_bounds) - 1, _stream()))

    def staged_item_slice(self, s, lr, reg_1, reg_2, gQ, cnt, loss_type=N.LOSS_BPR):
        check(lib.daisy_bpr_staged_item_slice(self._h, int(loss_type), _ptr(gQ, torch.float32, "gQ"),
                                              _ptr(cnt, torch.float32, "cnt"), int(s), float(lr), float(reg_1),
                                              float(reg_2), _ptr(self.stats, torch.float64, "stats"), _stream()))

    def item_apply_counts(self, Q_rows, g_rows, cnt_rows, lr, reg_1, reg_2):
        """The row owner's SGD step of a multi-GPU staged step: Q_rows -= lr*(g_rows + regulariser from the
        reduced entry counts and the finalized global norms); clears g_rows and cnt_rows."""
        item_apply_counts(Q_rows, g_rows, cnt_rows, lr, reg_1, reg_2, self.stats)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            check(lib.daisy_bpr_ctx_destroy(self._h))
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def scratch_bytes(self):
        return int(lib.daisy_bpr_ctx_scratch_bytes(self._h))

    def set_pointwise(self, flag):
        """Rows given to set_batch* are (user, item, label) - the CL / SL layout (sampler.py:93-98)."""
        check(lib.daisy_bpr_ctx_set_pointwise(self._h, int(bool(flag))))

    def set_bias(self, u_bias=None, i_bias=None, bias=None, g_u_bias=None, g_i_bias=None, g_bias=None):
        """Attach FM's bias parameters (FMRecommender.py:46-50); `set_bias()` detaches them.
        g_i_bias [I] is mandatory (zeroed scratch like gQ); g_u_bias [U] / g_bias [1] only for user_grad."""
        if u_bias is None:
            check(lib.daisy_bpr_ctx_set_bias(self._h, None, None, None, None, None, None))
            self._bias = None
            return
        f = torch.float32
        check(lib.daisy_bpr_ctx_set_bias(self._h, _ptr(u_bias.view(-1), f, "u_bias"), _ptr(i_bias.view(-1), f, "i_bias"),
                                         _ptr(bias.view(-1), f, "bias"), _ptr(g_u_bias, f, "g_u_bias"),
                                         _ptr(g_i_bias, f, "g_i_bias"), _ptr(g_bias, f, "g_bias")))
        self._bias = (u_bias, i_bias, bias, g_u_bias, g_i_bias, g_bias)     # keep the storage alive

    # -- batch -------------------------------------------------------------
    def set_batch_from_triples(self, triples, idx=None, start=0, B=None, user_base=0, validate=True):
        """validate: raise ValueError when an id lies outside the tables (one host sync; the reference raises
        IndexError in nn.Embedding).  Unvalidated out-of-range ids are treated as id 0, never dereferenced."""
        n = triples.shape[0]
        if B is None:
            B = idx.numel() if idx is not None else n - start
        check(lib.daisy_bpr_set_batch_from_triples(
            self._h, _ptr(triples, torch.int32, "triples"), n,
            _ptr(idx, torch.int64, "idx"), int(start), int(B), int(user_base), _stream()))
        if validate:
            check(lib.daisy_bpr_ctx_validate_batch(self._h, _stream()))

    def set_batch(self, u, i, j, validate=True):
        check(lib.daisy_bpr_set_batch(self._h, _ptr(u, torch.int32, "u"), _ptr(i, torch.int32, "i"),
                                      _ptr(j, torch.int32, "j"), u.numel(), _stream()))
        if validate:
            check(lib.daisy_bpr_ctx_validate_batch(self._h, _stream()))

    def set_batch_from_plan(self, plan, k):
        check(lib.daisy_bpr_set_batch_from_plan(self._h, plan._h, int(k), _stream()))

    # -- phases --------------------------------------------------------------
    def forward(self, P, Q, loss_type=N.LOSS_BPR, gamma=1e-10):
        check(lib.daisy_bpr_forward(self._h, _ptr(P, torch.float32, "P"), _ptr(Q, torch.float32, "Q"),
                                    int(loss_type), float(gamma),
                                    _ptr(self.stats, torch.float64, "stats"), _stream()))

    def finalize(self, reg_1, reg_2, step_loss=None, accumulate=True):
        check(lib.daisy_bpr_finalize(self._h, _ptr(self.stats, torch.float64, "stats"), float(reg_1),
                                     float(reg_2),
                                     _ptr(self.epoch_acc, torch.float64, "epoch_acc") if accumulate else None,
                                     _ptr(step_loss, torch.float64, "step_loss"), _stream()))

    def item_grad(self, P, Q, reg_1, reg_2, item_mode=N.ITEM_CHUNKED, gQ=None):
        gQ = self.gQ if gQ is None else gQ
        check(lib.daisy_bpr_item_grad(self._h, _ptr(P, torch.float32, "P"), _ptr(Q, torch.float32, "Q"),
                                      _ptr(self.stats, torch.float64, "stats"), float(reg_1),
                                      float(reg_2), _ptr(gQ, torch.float32, "gQ"), int(item_mode),
                                      _stream()))

    def item_grad_data(self, P, Q, item_mode=N.ITEM_CHUNKED, gQ=None):
        gQ = self.gQ if gQ is None else gQ
        check(lib.daisy_bpr_item_grad_data(self._h, _ptr(P, torch.float32, "P"), _ptr(Q, torch.float32, "Q"),
                                           _ptr(self.stats, torch.float64, "stats"),
                                           _ptr(gQ, torch.float32, "gQ"), int(item_mode), _stream()))

    def item_grad_reg(self, Q, reg_1, reg_2, gQ=None):
        gQ = self.gQ if gQ is None else gQ
        check(lib.daisy_bpr_item_grad_reg(self._h, _ptr(Q, torch.float32, "Q"),
                                          _ptr(self.stats, torch.float64, "stats"), float(reg_1),
                                          float(reg_2), _ptr(gQ, torch.float32, "gQ"), _stream()))

    def user_sgd(self, P, Q, lr, reg_1, reg_2):
        check(lib.daisy_bpr_user_sgd(self._h, _ptr(P, torch.float32, "P"), _ptr(Q, torch.float32, "Q"),
                                     _ptr(self.stats, torch.float64, "stats"), float(lr), float(reg_1),
                                     float(reg_2), _stream()))

    def user_grad(self, P, Q, reg_1, reg_2, gP):
        check(lib.daisy_bpr_user_grad(self._h, _ptr(P, torch.float32, "P"), _ptr(Q, torch.float32, "Q"),
                                      _ptr(self.stats, torch.float64, "stats"), float(reg_1),
                                      float(reg_2), _ptr(gP, torch.float32, "gP"), _stream()))

    def item_sgd_apply(self, Q, lr, dense=False, gQ=None):
        gQ = self.gQ if gQ is None else gQ
        check(lib.daisy_bpr_item_sgd_apply(self._h, _ptr(Q, torch.float32, "Q"),
                                           _ptr(gQ, torch.float32, "gQ"), float(lr), int(dense),
                                           _stream()))

    def sgd_step(self, P, Q, lr, reg_1, reg_2, loss_type=N.LOSS_BPR, gamma=1e-10,
                 item_mode=N.ITEM_CHUNKED, step_loss=None, accumulate=True):
        if item_mode == N.ITEM_FUSED:
            self._sync_norm_cache(P)
        check(lib.daisy_bpr_sgd_step(
            self._h, _ptr(P, torch.float32, "P"), _ptr(Q, torch.float32, "Q"), int(loss_type),
            float(gamma), float(lr), float(reg_1), float(reg_2), _ptr(self.gQ, torch.float32, "gQ"),
            _ptr(self.stats, torch.float64, "stats"),
            _ptr(self.epoch_acc, torch.float64, "epoch_acc") if accumulate else None,
            _ptr(step_loss, torch.float64, "step_loss"), int(item_mode), _stream()))

    def fit_epoch_sgd(self, plan, P, Q, lr, reg_1, reg_2, loss_type=N.LOSS_BPR, gamma=1e-10,
                      item_mode=N.ITEM_CHUNKED, step_losses=None):
        """Every batch of a built plan (one epoch), enqueued natively."""
        if item_mode == N.ITEM_FUSED:
            self._sync_norm_cache(P)
        check(lib.daisy_bpr_fit_epoch_sgd(
            self._h, plan._h, _ptr(P, torch.float32, "P"), _ptr(Q, torch.float32, "Q"),
            int(loss_type), float(gamma), float(lr), float(reg_1), float(reg_2),
            _ptr(self.gQ, torch.float32, "gQ"), _ptr(self.stats, torch.float64, "stats"),
            _ptr(self.epoch_acc, torch.float64, "epoch_acc"),
            _ptr(step_losses, torch.float64, "step_losses"), int(item_mode), _stream()))


def item_apply_counts(Q, g, cnt, lr, reg_1, reg_2, stats):
    """Row owner's SGD step from reduced (g, cnt) (multi-GPU staged step); clears g and cnt."""
    check(lib.daisy_item_apply_counts(_ptr(Q, torch.float32, "Q"), _ptr(g, torch.float32, "g"),
                                      _ptr(cnt, torch.float32, "cnt"), Q.shape[0], Q.shape[1], float(lr),
                                      float(reg_1), float(reg_2), _ptr(stats, torch.float64, "stats"), _stream()))


def adam_dense(W, g, m, v, lr, step, beta1=0.9, beta2=0.999, eps=1e-8):
    check(lib.daisy_adam_dense(_ptr(W, torch.float32, "W"), _ptr(g, torch.float32, "g"),
                               _ptr(m, torch.float32, "m"), _ptr(v, torch.float32, "v"), W.numel(),
                               float(lr), float(beta1), float(beta2), float(eps), int(step), _stream()))


def adagrad_dense(W, g, state_sum, lr, eps=1e-10):
    check(lib.daisy_adagrad_dense(_ptr(W, torch.float32, "W"), _ptr(g, torch.float32, "g"),
                                  _ptr(state_sum, torch.float32, "state_sum"), W.numel(), float(lr), float(eps), _stream()))


def rmsprop_dense(W, g, square_avg, lr, alpha=0.99, eps=1e-8):
    check(lib.daisy_rmsprop_dense(_ptr(W, torch.float32, "W"), _ptr(g, torch.float32, "g"),
                                  _ptr(square_avg, torch.float32, "square_avg"), W.numel(), float(lr), float(alpha),
                                  float(eps), _stream()))


class DenseOptimizer:
    """The dense torch optimisers of AbstractRecommender._build_optimizer (:48-67) on the native kernels, one
    state per parameter tensor: 'sgd', 'adam', 'adagrad', 'rmsprop' (torch defaults).  step() consumes and clears g."""

    KINDS = ("sgd", "adam", "adagrad", "rmsprop")

    def __init__(self, kind: str, lr: float):
        if kind not in self.KINDS:
            raise ValueError(f"DenseOptimizer: unknown kind {kind!r}")
        self.kind, self.lr, self.t, self._state = kind, float(lr), 0, {}

    def next_step(self):
        """Advance the (shared) step count: once per optimizer.step() of the reference."""
        self.t += 1

    def state_for(self, W):
        """The optimiser's state tensors of the flat parameter vector W (created zero at first use): () for SGD."""
        if self.kind == "sgd":
            return ()
        st = self._state.get(W.data_ptr())
        if st is None:
            st = self._state[W.data_ptr()] = tuple(torch.zeros_like(W) for _ in range(2 if self.kind == "adam" else 1))
        return st

    def step(self, W, g):
        W, g = W.view(-1), g.view(-1)
        if self.kind == "sgd":
            return sgd_dense(W, g, self.lr)
        st = self.state_for(W)
        if self.kind == "adam":
            adam_dense(W, g, st[0], st[1], self.lr, max(self.t, 1))
        elif self.kind == "adagrad":
            adagrad_dense(W, g, st[0], self.lr)
        else:
            rmsprop_dense(W, g, st[0], self.lr)


class LazyAdam:
    """torch.optim.Adam (defaults) on the two MF tables with traffic proportional to the rows a step touches, and the
    same bits as the dense form: rows without a gradient are replayed in registers when next needed
    (csrc/bpr_train.hip: k_adam_batch_rows / k_adam_flush).  Usage per step: `catchup(ctx)` before the forward
    pass, `step(ctx, gP, gQ)` after the gradients; `flush()` before anything else reads the tables."""

    BETA1, BETA2, EPS = 0.9, 0.999, 1e-8

    def __init__(self, P, Q, lr, max_steps):
        self.P, self.Q, self.lr, self.t = P, Q, float(lr), 0
        self.m = [torch.zeros_like(P), torch.zeros_like(Q)]
        self.v = [torch.zeros_like(P), torch.zeros_like(Q)]
        self.last = [torch.zeros(P.shape[0], dtype=torch.int32, device=P.device),
                     torch.zeros(Q.shape[0], dtype=torch.int32, device=Q.device)]
        self._table_steps = 0
        self._table = None
        self._grow(max(int(max_steps), 1))

    def _grow(self, n_steps):
        host = (C.c_float * (2 * (n_steps + 1)))()
        check(lib.daisy_adam_lazy_table(self.lr, self.BETA1, self.BETA2, n_steps, host))
        self._table = torch.frombuffer(host, dtype=torch.float32).clone().to(self.P.device)
        self._table_steps = n_steps

    def catchup(self, ctx):
        self.t += 1
        if self.t > self._table_steps:
            self._grow(2 * self.t)
        self._ctx = ctx            # (flush() rewrites rows of P behind this context's row-norm cache)
        ctx._p_key = None          # the native side drops the cache; so does the wrapper's notion of it
        f = torch.float32
        check(lib.daisy_adam_lazy_catchup(
            ctx._h, _ptr(self.P, f, "P"), _ptr(self.m[0], f, "mP"), _ptr(self.v[0], f, "vP"),
            _ptr(self.last[0], torch.int32, "lastP"), _ptr(self.Q, f, "Q"), _ptr(self.m[1], f, "mQ"),
            _ptr(self.v[1], f, "vQ"), _ptr(self.last[1], torch.int32, "lastQ"), _ptr(self._table, f, "table"),
            self.BETA1, self.BETA2, self.EPS, self.t, _stream()))

    def step(self, ctx, gP, gQ):
        self._ctx = ctx
        ctx._p_key = None
        f = torch.float32
        check(lib.daisy_adam_lazy_step(
            ctx._h, _ptr(self.P, f, "P"), _ptr(gP, f, "gP"), _ptr(self.m[0], f, "mP"), _ptr(self.v[0], f, "vP"),
            _ptr(self.last[0], torch.int32, "lastP"), _ptr(self.Q, f, "Q"), _ptr(gQ, f, "gQ"), _ptr(self.m[1], f, "mQ"),
            _ptr(self.v[1], f, "vQ"), _ptr(self.last[1], torch.int32, "lastQ"), _ptr(self._table, f, "table"),
            self.BETA1, self.BETA2, self.EPS, self.t, _stream()))

    def staged_step(self, ctx, reg_1, reg_2, loss_type=N.LOSS_BPR, gamma=1e-10, step_loss=None, accumulate=True):
        """One Adam step on the current batch through the staged kernels (forward fused into the user update, item rows
        committed by their segment owner): catch-up of the rows the batch references, then step t on the rows that
        have a gradient.  FM contexts: the bias gradients land in the context's g_u_bias / g_i_bias and
        stats[ST_SUM_COEF] (bias_) for the caller's dense optimiser."""
        self.t += 1
        if self.t > self._table_steps:
            self._grow(2 * self.t)
        ctx._sync_norm_cache(self.P)
        self._ctx = ctx
        f = torch.float32
        check(lib.daisy_bpr_staged_adam_step(
            ctx._h, _ptr(self.P, f, "P"), _ptr(self.Q, f, "Q"), int(loss_type), float(gamma), self.lr, float(reg_1),
            float(reg_2), _ptr(self.m[0], f, "mP"), _ptr(self.v[0], f, "vP"), _ptr(self.last[0], torch.int32, "lastP"),
            _ptr(self.m[1], f, "mQ"), _ptr(self.v[1], f, "vQ"), _ptr(self.last[1], torch.int32, "lastQ"),
            _ptr(self._table, f, "table"), self.BETA1, self.BETA2, self.EPS, self.t,
            _ptr(ctx.stats, torch.float64, "stats"),
            _ptr(ctx.epoch_acc, torch.float64, "epoch_acc") if accumulate else None,
            _ptr(step_loss, torch.float64, "step_loss"), _stream()))

    @staticmethod
    def small_epoch_supported(ctx, plan, loss_type=N.LOSS_BPR) -> bool:
        """True when `fit_epoch` runs this (sorted-layout) plan's epochs inside one persistent workgroup
        (daisy_bpr_small_epoch_supported: B <= 256, pairwise loss, no FM biases, rows that fit the LDS)"""
        return bool(lib.daisy_bpr_small_epoch_supported(ctx._h, plan._h, int(loss_type)) & 1)

    @staticmethod
    def small_epoch_pays(ctx, plan, loss_type=N.LOSS_BPR) -> bool:
        """... and the Adam form is expected to be faster than the chain of launches (small tables: the rows a step references
        sat out a few steps; measured 35 against 48 us per step at ml-100k shapes, but 750 against 250 at 1 M x 100 K tables)"""
        return lib.daisy_bpr_small_epoch_supported(ctx._h, plan._h, int(loss_type)) == 3

    def fit_epoch(self, ctx, plan, reg_1, reg_2, loss_type=N.LOSS_BPR, gamma=1e-10, flush=True, step_losses=None):
        """Every batch of a built plan through the staged Adam step, enqueued by ONE native call (daisy_bpr_fit_epoch_adam:
        the loop of AbstractRecommender.py:118-128 without a host round trip per batch), then - flush - the rows no batch
        referenced brought up to the epoch's last step.  MF contexts only (FM's biases step between two steps)."""
        nb = plan.num_batches
        if self.t + nb > self._table_steps:
            self._grow(2 * (self.t + nb))
        ctx._sync_norm_cache(self.P)
        self._ctx = ctx
        f = torch.float32
        check(lib.daisy_bpr_fit_epoch_adam(
            ctx._h, plan._h, _ptr(self.P, f, "P"), _ptr(self.Q, f, "Q"), int(loss_type), float(gamma), self.lr,
            float(reg_1), float(reg_2), _ptr(self.m[0], f, "mP"), _ptr(self.v[0], f, "vP"),
            _ptr(self.last[0], torch.int32, "lastP"), _ptr(self.m[1], f, "mQ"), _ptr(self.v[1], f, "vQ"),
            _ptr(self.last[1], torch.int32, "lastQ"), _ptr(self._table, f, "table"), self._table_steps, self.BETA1,
            self.BETA2, self.EPS, self.t + 1, 1 if flush else 0, _ptr(ctx.stats, torch.float64, "stats"),
            _ptr(ctx.epoch_acc, torch.float64, "epoch_acc"), _ptr(step_losses, torch.float64, "step_losses"), _stream()))
        self.t += nb

    def flush(self):
        f = torch.float32
        for W, m, v, last in ((self.P, self.m[0], self.v[0], self.last[0]), (self.Q, self.m[1], self.v[1], self.last[1])):
            check(lib.daisy_adam_lazy_flush(_ptr(W, f, "W"), _ptr(m, f, "m"), _ptr(v, f, "v"),
                                            _ptr(last, torch.int32, "last"), W.shape[0], W.shape[1],
                                            _ptr(self._table, f, "table"), self.BETA1, self.BETA2, self.EPS, self.t,
                                            _stream()))
        ctx = getattr(self, "_ctx", None)              # the flush rewrote rows of P behind the staged step's row-norm cache
        if ctx is not None and getattr(ctx, "_h", None) is not None and ctx._h.value:
            ctx.invalidate_cache()
        else:                                          # (a context that was closed in the meantime has no cache to drop)
            self._ctx = None


class ShardedAdam:
    """torch.optim.Adam for the user-sharded step (sharding.UserShardedBprTrainer): the rank's rows of P in the lazy form
    (moments + stamps per local user; catch-up of a batch's rows before its forward, flush at the end), the rank's OWN
    block(s) of Q densely (every owned row steps in every step, like torch).  Same expressions as daisy_adam_dense."""

    BETA1, BETA2, EPS = 0.9, 0.999, 1e-8

    def __init__(self, P_local, owned_q_rows, d, lr, max_steps):
        dev = P_local.device
        self.P, self.lr, self.t = P_local, float(lr), 0
        self.mP, self.vP = torch.zeros_like(P_local), torch.zeros_like(P_local)
        self.lastP = torch.zeros(P_local.shape[0], dtype=torch.int32, device=dev)
        self.mQ = torch.zeros(owned_q_rows, d, dtype=torch.float32, device=dev)
        self.vQ = torch.zeros(owned_q_rows, d, dtype=torch.float32, device=dev)
        self._steps = 0
        self.table = None
        self._grow(max(int(max_steps), 1))

    def _grow(self, n_steps):
        host = (C.c_float * (2 * (n_steps + 1)))()
        check(lib.daisy_adam_lazy_table(self.lr, self.BETA1, self.BETA2, n_steps, host))
        self.table = torch.frombuffer(host, dtype=torch.float32).clone().to(self.P.device)
        self._steps = n_steps

    def next_step(self):
        self.t += 1
        if self.t > self._steps:
            self._grow(2 * self.t)

    def flush(self, ctx=None):
        """every local row of P up to the current step"""
        if self.t == 0 or self.P.shape[0] == 0:
            return
        f = torch.float32
        check(lib.daisy_adam_lazy_flush(_ptr(self.P, f, "W"), _ptr(self.mP, f, "m"), _ptr(self.vP, f, "v"),
                                        _ptr(self.lastP, torch.int32, "last"), self.P.shape[0], self.P.shape[1],
                                        _ptr(self.table, f, "table"), self.BETA1, self.BETA2, self.EPS, self.t, _stream()))
        if ctx is not None:
            ctx.invalidate_cache()


def _bias_ptrs(biases):
    if biases is None:
        return None, None, None
    bu, bi, b0 = biases
    f = torch.float32
    return _ptr(bu.view(-1), f, "u_bias"), _ptr(bi.view(-1), f, "i_bias"), _ptr(b0.view(-1), f, "bias")


def mf_predict(P, Q, u, i, biases=None):
    """MF.forward (MFRecommender.py:63-68); biases=(u_bias, i_bias, bias_): FM.forward (FMRecommender.py:61-68)."""
    u = u.to(torch.int64).contiguous()
    i = i.to(torch.int64).contiguous()
    out = torch.empty(u.numel(), dtype=torch.float32, device=P.device)
    if u.numel() == 0:
        return out
    bu, bi, b0 = _bias_ptrs(biases)
    check(lib.daisy_fm_predict(_ptr(P, torch.float32, "P"), _ptr(Q, torch.float32, "Q"), bu, bi, b0, P.shape[1],
                               _ptr(u, torch.int64, "u"), _ptr(i, torch.int64, "i"), u.numel(),
                               _ptr(out, torch.float32, "out"), _stream()))
    return out.view(u.shape)


def mf_rank_topk(P, Q, us, cands, topk, return_scores=False, biases=None):
    """One batch of MF.rank (MFRecommender.py:109-121) -> int64 [B, topk]; biases: FM.rank
    (FMRecommender.py:105-123)."""
    us = us.to(torch.int64).contiguous()
    cands = cands.to(torch.int64).contiguous()
    B, Cn = cands.shape
    topk = min(int(topk), int(Cn))          # rank_list[:, :topk] truncates (MFRecommender.py:119)
    out = torch.empty(B, topk, dtype=torch.int64, device=P.device)
    scores = torch.empty(B, Cn, dtype=torch.float32, device=P.device) if return_scores else None
    nbytes = lib.daisy_mf_rank_workspace_bytes(B, Cn)
    ws = _ws(nbytes, P.device)
    bu, bi, b0 = _bias_ptrs(biases)
    check(lib.daisy_fm_rank_topk(_ptr(P, torch.float32, "P"), _ptr(Q, torch.float32, "Q"), bu, bi, b0, P.shape[1],
                                 _ptr(us, torch.int64, "us"), _ptr(cands, torch.int64, "cands"), B, Cn,
                                 int(topk), _ptr(out, torch.int64, "out"),
                                 _ptr(scores, torch.float32, "scores"), _ptr(ws, torch.uint8, "ws"),
                                 ws.numel(), _stream()))
    return (out, scores) if return_scores else out


def mf_full_rank(P, Q, u, topk, biases=None):
    """MF.full_rank (MFRecommender.py:126-133) -> int64 [topk]; biases: FM.full_rank (FMRecommender.py:125-133)."""
    I = Q.shape[0]
    topk = min(int(topk), int(I))           # argsort(...)[:topk] truncates (MFRecommender.py:131)
    out = torch.empty(topk, dtype=torch.int64, device=P.device)
    ws = _ws(lib.daisy_mf_full_rank_workspace_bytes(I), P.device)
    bu, bi, b0 = _bias_ptrs(biases)
    check(lib.daisy_fm_full_rank(_ptr(P, torch.float32, "P"), _ptr(Q, torch.float32, "Q"), bu, bi, b0, P.shape[1], I,
                                 int(u), int(topk), _ptr(out, torch.int64, "out"),
                                 _ptr(ws, torch.uint8, "ws"), ws.numel(), _stream()))
    return out


def build_user_csr(users, items, user_num):
    """get_ur (utils.py:19-34) as a device CSR: (indptr int64[U+1], sorted items int32[n])."""
    n = users.numel()
    indptr = torch.empty(user_num + 1, dtype=torch.int64, device=users.device)
    csr_items = torch.empty(n, dtype=torch.int32, device=users.device)
    ws = _ws(lib.daisy_csr_workspace_bytes(n), users.device)
    check(lib.daisy_build_user_csr(_ptr(users, torch.int32, "users"), _ptr(items, torch.int32, "items"),
                                   n, int(user_num), _ptr(indptr, torch.int64, "indptr"),
                                   _ptr(csr_items, torch.int32, "csr_items"), _ptr(ws, torch.uint8, "ws"),
                                   ws.numel(), _stream()))
    return indptr, csr_items


def sample_neg_per_user(indptr, csr_items, item_num, num_ng, seed, epoch=0):
    U = indptr.numel() - 1
    js = torch.empty(U, num_ng, dtype=torch.int32, device=indptr.device)
    check(lib.daisy_sample_neg_per_user(_ptr(indptr, torch.int64, "indptr"),
                                        _ptr(csr_items, torch.int32, "csr_items"), U, int(item_num),
                                        int(num_ng), int(seed), int(epoch), _ptr(js, torch.int32, "js"),
                                        _stream()))
    return js


POP_STREAM = 1 << 62          # Philox stream of the popularity-weighted draws (the uniform ones use `epoch`)


def sample_categorical(cdf, rows, k, seed, stream_id, out=None, col0=0):
    """k draws per row from the categorical distribution with inclusive cumulative sums `cdf` (float64 [I], device):
    the 'high-pop' / 'low-pop' share of sampler.py:76-80.  Written into out[:, col0:col0+k] (int32 [rows, ld])."""
    if out is None:
        out = torch.empty(rows, k, dtype=torch.int32, device=cdf.device)
    check(lib.daisy_sample_categorical(_ptr(cdf, torch.float64, "cdf"), cdf.numel(), int(rows), int(k), int(seed),
                                       int(stream_id), _ptr(out, torch.int32, "out"), out.shape[1], int(col0), _stream()))
    return out


def skipgram_samples(seq_items, seq_user, seq_ptr, context_window, ur_indptr, ur_items, item_num, seed, stream_id=0):
    """SkipGramNegativeSampler.sampling (sampler.py:133-155) on the device: int32 [rows, 3] (target, context, label)
    rows, element by element of the user sequences - window positives, then as many uniform negatives."""
    dev = seq_items.device
    n = seq_items.numel()
    pos = torch.arange(n, device=dev, dtype=torch.int64) - seq_ptr[seq_user.long()]
    length = (seq_ptr[1:] - seq_ptr[:-1])[seq_user.long()]
    w = int(context_window)
    cnt = torch.clamp(pos, max=w) + torch.clamp(length - 1 - pos, max=w)                 # window size of every element
    off = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    off[1:] = torch.cumsum(2 * cnt, 0)
    rows = int(off[-1].item())
    out = torch.empty(rows, 3, dtype=torch.int32, device=dev)
    bad = torch.zeros(1, dtype=torch.int32, device=dev)
    if rows:
        check(lib.daisy_skipgram_samples(_ptr(seq_items, torch.int32, "seq_items"), _ptr(seq_user, torch.int32, "seq_user"),
                                         _ptr(seq_ptr, torch.int64, "seq_ptr"), _ptr(off, torch.int64, "row_offsets"), n, w,
                                         _ptr(ur_indptr, torch.int64, "ur_indptr"), _ptr(ur_items, torch.int32, "ur_items"),
                                         int(item_num), int(seed), int(stream_id), _ptr(out, torch.int32, "out"),
                                         _ptr(bad, torch.int32, "bad"), _stream()))
    if int(bad.item()):
        raise ValueError("'a' cannot be empty unless no samples are taken")      # np.random.choice on an empty complement
    return out


def expand_triples(users, items, js):
    n, num_ng = users.numel(), js.shape[1]
    out = torch.empty(n * num_ng, 3, dtype=torch.int32, device=users.device)
    check(lib.daisy_expand_triples(_ptr(users, torch.int32, "users"), _ptr(items, torch.int32, "items"),
                                   n, _ptr(js, torch.int32, "js"), num_ng,
                                   _ptr(out, torch.int32, "triples"), _stream()))
    return out


def resample_neg_per_interaction(indptr, csr_items, item_num, triples, seed, epoch=0):
    check(lib.daisy_resample_neg_per_interaction(
        _ptr(indptr, torch.int64, "indptr"), _ptr(csr_items, torch.int32, "csr_items"), int(item_num),
        _ptr(triples, torch.int32, "triples"), triples.shape[0], int(seed), int(epoch), _stream()))
    return triples


def build_candidates(indptr_test, items_test, indptr_train, items_train, users, item_num, cand_num, seed):
    """build_candidates_set (utils.py:53-85) on the device -> int64 [n_users, cand_num]."""
    users = users.to(torch.int64).contiguous()
    out = torch.empty(users.numel(), cand_num, dtype=torch.int64, device=users.device)
    check(lib.daisy_build_candidates(_ptr(indptr_test, torch.int64, "indptr_test"),
                                     _ptr(items_test, torch.int32, "items_test"),
                                     _ptr(indptr_train, torch.int64, "indptr_train"),
                                     _ptr(items_train, torch.int32, "items_train"),
                                     _ptr(users, torch.int64, "users"), users.numel(), int(item_num),
                                     int(cand_num), int(seed), _ptr(out, torch.int64, "out"), _stream()))
    return out


def randperm(n, seed, epoch=0, device="cuda"):
    perm = torch.empty(n, dtype=torch.int64, device=device)
    ws = _ws(lib.daisy_randperm_workspace_bytes(n), perm.device)
    check(lib.daisy_randperm(int(n), int(seed), int(epoch), _ptr(perm, torch.int64, "perm"),
                             _ptr(ws, torch.uint8, "ws"), ws.numel(), _stream()))
    return perm


def membench(what, table, idx, out):
    check(lib.daisy_membench(int(what), _ptr(table, torch.float32, "table"), table.shape[0],
                             table.shape[1], _ptr(idx, torch.int32, "idx"), idx.numel(),
                             _ptr(out, torch.float32, "out"), _stream()))


# ------------------------------------------------------------------------------------------------
# NeuMF (NeuMFRecommender.py) - see include/daisyrec_amd.h
# ------------------------------------------------------------------------------------------------
NEUMF_MODELS = {"NeuMF": N.NEUMF_FULL, "NeuMF-end": N.NEUMF_FULL, "GMF": N.NEUMF_GMF, "MLP": N.NEUMF_MLP}


def neumf_param_names(num_layers):
    names = ["uG", "iG", "uM", "iM"]
    for l in range(1, num_layers + 1):
        names += [f"W{l}", f"b{l}"]
    return names + ["Wp", "bp"]


def _neumf_table(tensors, num_layers):
    """dict name -> float32 device tensor  =>  daisy_neumf_params"""
    t = N.NeumfParams()
    f = torch.float32
    for k in ("uG", "iG", "uM", "iM", "Wp", "bp"):
        setattr(t, k, _ptr(tensors[k], f, k))
    for l in range(1, num_layers + 1):
        t.W[l - 1] = _ptr(tensors[f"W{l}"], f, f"W{l}")
        t.b[l - 1] = _ptr(tensors[f"b{l}"], f, f"b{l}")
    return t


class NeumfContext:
    """Activation workspace + entry points of the NeuMF path (daisy_neumf_*)."""

    def __init__(self, max_rows, factors, num_layers, user_num, item_num, model="NeuMF", device=None):
        if model not in NEUMF_MODELS:
            raise NotImplementedError(f"NeuMF model_name '{model}' (native: {sorted(NEUMF_MODELS)})")
        self.device = torch.device(device if device is not None else "cuda")
        self.L, self.d = int(num_layers), int(factors)
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            check(lib.daisy_neumf_ctx_create(C.byref(self._h), int(max_rows), int(factors), int(num_layers),
                                             NEUMF_MODELS[model], int(user_num), int(item_num)))
        self.max_rows = int(max_rows)
        self.stats = torch.zeros(N.NEUMF_STATS_LEN, dtype=torch.float64, device=self.device)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            lib.daisy_neumf_ctx_destroy(self._h)
            self._h = None

    __del__ = close

    @property
    def nbytes(self):
        return int(lib.daisy_neumf_ctx_bytes(self._h))

    def set_precision(self, bf16_gemm):
        """0 / False (default): exact fp32; 1 / True: bf16-input MFMA in the MLP tower (fp32 operands rounded on the way
        into LDS); 2: activations and a copy of the weights stored as bf16 in HBM too (throughput modes)."""
        check(lib.daisy_neumf_ctx_set_precision(self._h, int(bf16_gemm)))

    def scores(self, params, users, items=None, C_=0, n=None):
        """NeuMF.forward in eval mode: see daisy_neumf_scores for the three pair layouts."""
        users = users.to(torch.int64).contiguous()
        if items is not None:
            items = items.to(torch.int64).contiguous()
            n = items.numel()
        out = torch.empty(int(n), dtype=torch.float32, device=self.device)
        tab = _neumf_table(params, self.L)
        check(lib.daisy_neumf_scores(self._h, C.byref(tab), _ptr(users, torch.int64, "users"),
                                     _ptr(items, torch.int64, "items"), int(n), int(C_),
                                     _ptr(out, torch.float32, "out"), _stream()))
        return out

    def step_grads(self, params, grads, u, i, j, loss_type=N.LOSS_BPR, reg_1=0.0, reg_2=0.0, dropout=0.0,
                   seed=0, gamma=1e-10):
        """NeuMF.calc_loss + backward for one batch: accumulates into `grads`, loss in stats[NST_LOSS]."""
        pt, gt = _neumf_table(params, self.L), _neumf_table(grads, self.L)
        check(lib.daisy_neumf_step_grads(self._h, C.byref(pt), C.byref(gt), _ptr(u, torch.int32, "u"),
                                         _ptr(i, torch.int32, "i"), _ptr(j, torch.int32, "j"), u.numel(),
                                         int(loss_type), float(gamma), float(reg_1), float(reg_2), float(dropout),
                                         int(seed), _ptr(self.stats, torch.float64, "stats"), _stream()))


    def fit_epoch(self, params, grads, u, i, j, batch, optim, W, g, loss_type=N.LOSS_BPR, reg_1=0.0, reg_2=0.0, dropout=0.0,
                  seed_hi=0, step0=0, gamma=1e-10):
        """One epoch of AbstractRecommender.fit's loop (:112-128) over the batches of (u, i, j), issued by the library
        (daisy_neumf_fit_epoch): step_grads + the dense optimiser `optim` on the flat vectors W / g per batch.  Advances
        optim.t by the number of steps and returns that number; the epoch's loss accumulates in stats[NST_LOSS_SUM]."""
        n = int(u.numel())
        steps = (n + int(batch) - 1) // int(batch)
        if optim.t != int(step0):
            raise ValueError(f"fit_epoch: the optimiser has taken {optim.t} steps, step0 = {step0}")
        W, g = W.view(-1), g.view(-1)
        st = optim.state_for(W)
        pt, gt = _neumf_table(params, self.L), _neumf_table(grads, self.L)
        f = torch.float32
        check(lib.daisy_neumf_fit_epoch(self._h, C.byref(pt), C.byref(gt), _ptr(u, torch.int32, "u"), _ptr(i, torch.int32, "i"),
                                        _ptr(j, torch.int32, "j"), n, int(batch), int(loss_type), float(gamma), float(reg_1),
                                        float(reg_2), float(dropout), int(seed_hi), int(step0),
                                        DenseOptimizer.KINDS.index(optim.kind), float(optim.lr), _ptr(W, f, "W"), _ptr(g, f, "g"),
                                        _ptr(st[0], f, "state0") if len(st) > 0 else None,
                                        _ptr(st[1], f, "state1") if len(st) > 1 else None, W.numel(),
                                        _ptr(self.stats, torch.float64, "stats"), _stream()))
        optim.t += steps
        return steps


def sgd_dense(W, g, lr):
    check(lib.daisy_sgd_dense(_ptr(W, torch.float32, "W"), _ptr(g, torch.float32, "g"), W.numel(), float(lr),
                              _stream()))


def topk_from_scores(scores, cands, topk):
    """argsort(descending, stable) + gather + [:topk] of every rank() (e.g. NeuMFRecommender.py:203-206)."""
    cands = cands.to(torch.int64).contiguous()
    B, Cn = cands.shape
    out = torch.empty(B, topk, dtype=torch.int64, device=scores.device)
    ws = _ws(lib.daisy_mf_rank_workspace_bytes(B, Cn), scores.device)
    check(lib.daisy_topk_from_scores(_ptr(scores.contiguous(), torch.float32, "scores"),
                                     _ptr(cands, torch.int64, "cands"), B, Cn, int(topk),
                                     _ptr(out, torch.int64, "out"), _ptr(ws, torch.uint8, "ws"), ws.numel(),
                                     _stream()))
    return out


def full_topk_from_scores(scores, topk):
    I = scores.numel()
    out = torch.empty(topk, dtype=torch.int64, device=scores.device)
    ws = _ws(lib.daisy_mf_full_rank_workspace_bytes(I), scores.device)
    check(lib.daisy_full_topk_from_scores(_ptr(scores.contiguous(), torch.float32, "scores"), I, int(topk),
                                          _ptr(out, torch.int64, "out"), _ptr(ws, torch.uint8, "ws"), ws.numel(),
                                          _stream()))
    return out


def gemm_nt_bf16(A, B):
    """C = A @ B.T with bf16 storage for A [M,K], B [N,K] and C [M,N] (fp32 accumulation on the MFMA units)."""
    M, K = A.shape
    Nn = B.shape[0]
    out = torch.empty(M, Nn, dtype=torch.bfloat16, device=A.device)
    check(lib.daisy_gemm_nt_bf16(_ptr(A, torch.bfloat16, "A"), _ptr(B, torch.bfloat16, "B"), _ptr(out, torch.bfloat16, "C"),
                                 M, Nn, K, _stream()))
    return out


def gemm_tn_bf16(At, Bt, k_chunk=2048):
    """C = At.T @ Bt for bf16 At [K,M], Bt [K,N] (the weight-gradient layout of NeuMF's precision level 2), fp32
    result, the reduction cut into k_chunk slices (fp32 atomics)."""
    K, M = At.shape
    Nn = Bt.shape[1]
    out = torch.zeros(M, Nn, dtype=torch.float32, device=At.device)
    check(lib.daisy_gemm_tn_bf16(_ptr(At, torch.bfloat16, "At"), _ptr(Bt, torch.bfloat16, "Bt"),
                                 _ptr(out, torch.float32, "C"), M, Nn, K, int(k_chunk), _stream()))
    return out


def gemm_nt(A, B, bf16=False):
    """C = A @ B.T on the MFMA tile kernels of the NeuMF tower (test / bench hook)."""
    M, K = A.shape
    Nn = B.shape[0]
    out = torch.empty(M, Nn, dtype=torch.float32, device=A.device)
    check(lib.daisy_gemm_nt(_ptr(A, torch.float32, "A"), _ptr(B, torch.float32, "B"),
                            _ptr(out, torch.float32, "C"), M, Nn, K, int(bool(bf16)), _stream()))
    return out


# ------------------------------------------------------------------------------------------------
# LightGCN (LightGCNRecommender.py) - see include/daisyrec_amd.h
# ------------------------------------------------------------------------------------------------
class LgcnGraph:
    """Normalised adjacency A_hat = D^-1/2 A D^-1/2 of the user-item graph on the device
    (LightGCNRecommender.py:74-107) and the products built on it."""

    def __init__(self, users, items, user_num, item_num):
        users = users.to(torch.int32).contiguous()
        items = items.to(torch.int32).contiguous()
        self.device = users.device
        self.U, self.I = int(user_num), int(item_num)
        self.N = self.U + self.I
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            check(lib.daisy_lgcn_graph_create(C.byref(self._h), _ptr(users, torch.int32, "users"),
                                              _ptr(items, torch.int32, "items"), users.numel(), self.U, self.I,
                                              _stream()))
        self.nnz = int(lib.daisy_lgcn_graph_nnz(self._h))
        self._work = None

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            lib.daisy_lgcn_graph_destroy(self._h)
            self._h = None

    __del__ = close

    @property
    def nbytes(self):
        return int(lib.daisy_lgcn_graph_bytes(self._h))

    def set_reproducible(self, flag):
        """True: row-owner products (bitwise reproducible); False (default): chunked segmented reduction."""
        check(lib.daisy_lgcn_graph_set_reproducible(self._h, int(bool(flag))))

    def coo(self):
        row = torch.empty(self.nnz, dtype=torch.int32, device=self.device)
        col = torch.empty_like(row)
        val = torch.empty(self.nnz, dtype=torch.float32, device=self.device)
        check(lib.daisy_lgcn_graph_read(self._h, _ptr(row, torch.int32, "row"), _ptr(col, torch.int32, "col"),
                                        _ptr(val, torch.float32, "val"), _stream()))
        return row, col, val

    def _scratch(self, d):
        if self._work is None or self._work.numel() < 2 * self.N * d:
            self._work = torch.empty(2 * self.N * d, dtype=torch.float32, device=self.device)
        return self._work

    def spmm(self, X):
        Y = torch.empty_like(X)
        check(lib.daisy_lgcn_spmm(self._h, _ptr(X, torch.float32, "X"), _ptr(Y, torch.float32, "Y"), X.shape[1],
                                  _stream()))
        return Y

    def spmm_rows(self, X, Yrows, row_lo, row_hi):
        """Yrows[1 + r - row_lo] = (A_hat X)[r] for r in [row_lo, row_hi): `Yrows` is [row_hi-row_lo+2, d] - the
        block plus one spare row on either side (the segmented reduction may spill a partial neighbour row there)."""
        assert Yrows.shape[0] >= row_hi - row_lo + 2 and Yrows.is_contiguous()
        check(lib.daisy_lgcn_spmm_rows(self._h, _ptr(X, torch.float32, "X"), _ptr(Yrows[1:], torch.float32, "Yrows"),
                                       X.shape[1], int(row_lo), int(row_hi), _stream()))
        return Yrows[1:1 + row_hi - row_lo]

    def propagate(self, E0, num_layers, out=None):
        """LightGCN.forward (LightGCNRecommender.py:117-129): mean_k A_hat^k E0, [N, d]."""
        out = torch.empty_like(E0) if out is None else out
        w = self._scratch(E0.shape[1])
        check(lib.daisy_lgcn_propagate(self._h, _ptr(E0, torch.float32, "E0"), E0.shape[1], int(num_layers),
                                       _ptr(w, torch.float32, "work"), _ptr(out, torch.float32, "out"), _stream()))
        return out

    def backprop(self, G, num_layers, dE0):
        """dE0 += 1/(L+1) sum_k A_hat^k G (the transpose of propagate; A_hat is symmetric)."""
        w = self._scratch(G.shape[1])
        check(lib.daisy_lgcn_backprop(self._h, _ptr(G, torch.float32, "G"), G.shape[1], int(num_layers),
                                      _ptr(w, torch.float32, "work"), _ptr(dE0, torch.float32, "dE0"), _stream()))


_LGCN_REG_WS = {}


def lgcn_reg_grad(E0, u, i, j, user_num, pointwise, reg_1, reg_2, stats, dE0):
    N = E0.shape[0]
    key = (E0.device, N)
    ws = _LGCN_REG_WS.get(key)            # int32 occurrence counts, kept all-zero between calls by the kernels
    if ws is None:
        ws = _LGCN_REG_WS[key] = torch.zeros(2 * N, dtype=torch.int32, device=E0.device)
    check(lib.daisy_lgcn_reg_grad(_ptr(E0, torch.float32, "E0"), _ptr(u, torch.int32, "u"), _ptr(i, torch.int32, "i"),
                                  _ptr(j, torch.int32, "j"), u.numel(), int(user_num), int(N - user_num), E0.shape[1],
                                  int(bool(pointwise)), float(reg_1), float(reg_2), _ptr(stats, torch.float64, "stats"),
                                  _ptr(ws, torch.int32, "count_ws"), _ptr(dE0, torch.float32, "dE0"), _stream()))


def axpby(x, a, b, y, zero_x=False):
    """y = a*x + b*y (and x = 0 when zero_x)."""
    check(lib.daisy_axpby_f32(_ptr(x, torch.float32, "x"), float(a), float(b), _ptr(y, torch.float32, "y"), y.numel(),
                              int(bool(zero_x)), _stream()))


def csr_row_sum(indptr, cols, X, out):
    """out[r] = sum of X[cols[e]] over the CSR row r (rows without entries untouched)."""
    check(lib.daisy_csr_row_sum(_ptr(indptr, torch.int64, "indptr"), _ptr(cols, torch.int32, "cols"),
                                _ptr(X, torch.float32, "X"), indptr.numel() - 1, X.shape[1],
                                _ptr(out, torch.float32, "out"), _stream()))
