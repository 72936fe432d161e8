"""ctypes binding of ``libdaisyrec_hip.so`` (declared in ``include/daisyrec_amd.h``).

The HIP library IS the product: there is no CPU or PyTorch fallback.  Importing
this module without the built library raises ``ImportError`` with the build
command, and every call checks the status code and raises.
"""
from __future__ import annotations

import ctypes as C
import os

# torch bundles its own HIP/HSA runtime (torch/lib/libamdhip64.so, SONAME libamdhip64.so.7).
# It MUST be mapped before our library so that the dynamic linker binds our NEEDED
# libamdhip64.so.7 to that same copy: two HIP runtimes in one process do not share streams
# or device state (observed: "no ROCm-capable device is detected" from the second copy).
import torch  # noqa: F401  (side effect: loads torch's HIP runtime first)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DAISY_LIB_OVERRIDE") or os.path.join(_HERE, "lib", "libdaisyrec_hip.so")

DAISY_OK, DAISY_ERR_ARG, DAISY_ERR_HIP, DAISY_ERR_STATE = 0, 1, 2, 3
LOSS_BPR, LOSS_HL, LOSS_TL, LOSS_CL, LOSS_SL = 0, 1, 2, 3, 4
ITEM_ATOMIC, ITEM_SORTED, ITEM_CHUNKED, ITEM_FUSED = 0, 1, 2, 3
ORDER_IDENTITY, ORDER_PERM, ORDER_FEISTEL = 0, 1, 2
PLAN_TRIPLES_USER_SORTED = 1
PLAN_POINTWISE = 2
STATS_LEN = 16
ST_LOSS_DATA, ST_L1_U, ST_L1_I, ST_L1_J, ST_SQ_U, ST_SQ_I, ST_SQ_J = range(7)
ST_LOSS, ST_NORM_U, ST_NORM_I, ST_NORM_J = 7, 8, 9, 10
ST_SUM_COEF = 12
ST_SQ_U_PRE = 13
ABI_VERSION = 7

_p = C.c_void_p
_i32, _i64, _u64, _f32, _sz = C.c_int32, C.c_int64, C.c_uint64, C.c_float, C.c_size_t

NEUMF_MAX_LAYERS = 8
NEUMF_FULL, NEUMF_GMF, NEUMF_MLP = 0, 1, 2
NST_LOSS_DATA, NST_L1, NST_SQ, NST_LOSS, NST_NORM, NST_LOSS_SUM, NEUMF_STATS_LEN = 0, 1, 6, 11, 12, 17, 24


class NeumfParams(C.Structure):
    """daisy_neumf_params: table of device pointers (include/daisyrec_amd.h)."""
    _fields_ = [("uG", _p), ("iG", _p), ("uM", _p), ("iM", _p),
                ("W", _p * NEUMF_MAX_LAYERS), ("b", _p * NEUMF_MAX_LAYERS), ("Wp", _p), ("bp", _p)]


_pp = C.POINTER(NeumfParams)

# name -> (restype, argtypes); every symbol include/daisyrec_amd.h declares
SIGNATURES = {
    "daisy_last_error": (C.c_char_p, []),
    "daisy_abi_version": (C.c_int, []),
    "daisy_bpr_ctx_create": (C.c_int, [C.POINTER(_p), _i64, _i32, _i64, _i64]),
    "daisy_bpr_ctx_destroy": (C.c_int, [_p]),
    "daisy_bpr_ctx_scratch_bytes": (_sz, [_p]),
    "daisy_bpr_set_batch_from_triples": (C.c_int, [_p, _p, _i64, _p, _i64, _i64, _i32, _p]),
    "daisy_bpr_set_batch": (C.c_int, [_p, _p, _p, _p, _i64, _p]),
    "daisy_bpr_set_batch_from_plan": (C.c_int, [_p, _p, _i64, _p]),
    "daisy_bpr_ctx_set_pointwise": (C.c_int, [_p, _i32]),
    "daisy_bpr_ctx_set_bias": (C.c_int, [_p, _p, _p, _p, _p, _p, _p]),
    "daisy_epoch_plan_create": (C.c_int, [C.POINTER(_p), _i64, _i64, _i64]),
    "daisy_epoch_plan_destroy": (C.c_int, [_p]),
    "daisy_epoch_plan_bytes": (_sz, [_p]),
    "daisy_epoch_plan_build": (C.c_int, [_p, _p, _i64, _p, _i32, _u64, _u64, _i64, _i32, _i32, _p]),
    "daisy_epoch_plan_num_batches": (_i64, [_p]),
    "daisy_epoch_plan_validate": (C.c_int, [_p, _p]),
    "daisy_bpr_ctx_validate_batch": (C.c_int, [_p, _p]),
    "daisy_epoch_plan_read_batch": (C.c_int, [_p, _i64, _p, _p, _p, _p, _p, _p, C.POINTER(_i64), _p]),
    "daisy_feistel_positions": (C.c_int, [_i64, _u64, _u64, _p, _p]),
    "daisy_feistel_positions_at": (C.c_int, [_p, _i64, _i64, _u64, _u64, _p, _p]),
    "daisy_train_index_create": (C.c_int, [C.POINTER(_p), _p, _i64, _i64, _i64, _i32, _i32, _p]),
    "daisy_train_index_destroy": (C.c_int, [_p]),
    "daisy_train_index_bytes": (_sz, [_p]),
    "daisy_epoch_plan_build_indexed": (C.c_int, [_p, _p, _p, _i32, _u64, _u64, _i64, _p]),
    "daisy_epoch_plan_build_positions": (C.c_int, [_p, _p, _p, _i64, _i64, _p]),
    "daisy_epoch_plan_batch_rows": (_i64, [_p, _i64]),
    "daisy_bpr_ctx_invalidate_cache": (C.c_int, [_p]),
    "daisy_bpr_ctx_set_p_stream": (C.c_int, [_p, C.c_int32]),
    "daisy_bpr_staged_prenorm": (C.c_int, [_p, _p, _p, _p]),
    "daisy_bpr_staged_user": (C.c_int, [_p, _p, _p, _i32, _f32, _f32, _f32, _f32, _p, _p]),
    "daisy_bpr_staged_item": (C.c_int, [_p, _i32, _p, _p, _p, _f32, _f32, _f32, _p, _p]),
    "daisy_item_apply_counts": (C.c_int, [_p, _p, _p, _i64, _i32, _f32, _f32, _f32, _p, _p]),
    "daisy_bpr_staged_item_slices": (C.c_int, [_p, _p, _i32, _p]),
    "daisy_bpr_staged_item_slice": (C.c_int, [_p, _i32, _p, _p, _i32, _f32, _f32, _f32, _p, _p]),
    "daisy_bpr_staged_adam_step": (C.c_int, [_p, _p, _p, _i32, _f32, _f32, _f32, _f32, _p, _p, _p, _p, _p, _p, _p,
                                             _f32, _f32, _f32, _i64, _p, _p, _p, _p]),
    "daisy_bpr_staged_adam_catchup_users": (C.c_int, [_p, _p, _p, _p, _p, _p, _f32, _f32, _f32, _i64, _p]),
    "daisy_bpr_staged_user_adam": (C.c_int, [_p, _p, _p, _i32, _f32, _f32, _f32, _f32, _p, _p, _p, _f32, _f32, _f32, _i64,
                                             _p, _p]),
    "daisy_item_apply_counts_adam": (C.c_int, [_p, _p, _p, _p, _p, _i64, _i32, _f32, _f32, _f32, _f32, _f32, _f32, _i64,
                                               _p, _p]),
    "daisy_bpr_forward": (C.c_int, [_p, _p, _p, _i32, _f32, _p, _p]),
    "daisy_bpr_finalize": (C.c_int, [_p, _p, _f32, _f32, _p, _p, _p]),
    "daisy_bpr_item_grad": (C.c_int, [_p, _p, _p, _p, _f32, _f32, _p, _i32, _p]),
    "daisy_bpr_item_grad_data": (C.c_int, [_p, _p, _p, _p, _p, _i32, _p]),
    "daisy_bpr_item_grad_reg": (C.c_int, [_p, _p, _p, _f32, _f32, _p, _p]),
    "daisy_bpr_user_sgd": (C.c_int, [_p, _p, _p, _p, _f32, _f32, _f32, _p]),
    "daisy_bpr_user_grad": (C.c_int, [_p, _p, _p, _p, _f32, _f32, _p, _p]),
    "daisy_bpr_item_sgd_apply": (C.c_int, [_p, _p, _p, _f32, _i32, _p]),
    "daisy_adam_dense": (C.c_int, [_p, _p, _p, _p, _i64, _f32, _f32, _f32, _f32, _i64, _p]),
    "daisy_adam_lazy_table": (C.c_int, [_f32, _f32, _f32, _i64, _p]),
    "daisy_adam_lazy_catchup": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _f32, _f32, _f32, _i64, _p]),
    "daisy_adam_lazy_step": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _f32, _f32, _f32, _i64, _p]),
    "daisy_adam_lazy_flush": (C.c_int, [_p, _p, _p, _p, _i64, _i32, _p, _f32, _f32, _f32, _i64, _p]),
    "daisy_adagrad_dense": (C.c_int, [_p, _p, _p, _i64, _f32, _f32, _p]),
    "daisy_rmsprop_dense": (C.c_int, [_p, _p, _p, _i64, _f32, _f32, _f32, _p]),
    "daisy_bpr_sgd_step": (C.c_int, [_p, _p, _p, _i32, _f32, _f32, _f32, _f32, _p, _p, _p, _p,
                                     _i32, _p]),
    "daisy_bpr_fit_epoch_sgd": (C.c_int, [_p, _p, _p, _p, _i32, _f32, _f32, _f32, _f32, _p, _p, _p,
                                          _p, _i32, _p]),
    "daisy_bpr_fit_epoch_adam": (C.c_int, [_p, _p, _p, _p, _i32, _f32, _f32, _f32, _f32, _p, _p, _p, _p, _p, _p, _p, _i64,
                                           _f32, _f32, _f32, _i64, _i32, _p, _p, _p, _p]),
    "daisy_bpr_small_epoch_supported": (C.c_int, [_p, _p, _i32]),
    "daisy_mf_predict": (C.c_int, [_p, _p, _i32, _p, _p, _i64, _p, _p]),
    "daisy_mf_rank_workspace_bytes": (_sz, [_i64, _i64]),
    "daisy_mf_rank_topk": (C.c_int, [_p, _p, _i32, _p, _p, _i64, _i64, _i32, _p, _p, _p, _sz, _p]),
    "daisy_mf_full_rank_workspace_bytes": (_sz, [_i64]),
    "daisy_mf_full_rank": (C.c_int, [_p, _p, _i32, _i64, _i64, _i32, _p, _p, _sz, _p]),
    "daisy_fm_predict": (C.c_int, [_p, _p, _p, _p, _p, _i32, _p, _p, _i64, _p, _p]),
    "daisy_fm_rank_topk": (C.c_int, [_p, _p, _p, _p, _p, _i32, _p, _p, _i64, _i64, _i32, _p, _p, _p, _sz, _p]),
    "daisy_fm_full_rank": (C.c_int, [_p, _p, _p, _p, _p, _i32, _i64, _i64, _i32, _p, _p, _sz, _p]),
    "daisy_neumf_ctx_create": (C.c_int, [C.POINTER(_p), _i64, _i32, _i32, _i32, _i64, _i64]),
    "daisy_neumf_ctx_destroy": (C.c_int, [_p]),
    "daisy_neumf_ctx_bytes": (_sz, [_p]),
    "daisy_neumf_ctx_set_precision": (C.c_int, [_p, _i32]),
    "daisy_neumf_scores": (C.c_int, [_p, _pp, _p, _p, _i64, _i64, _p, _p]),
    "daisy_neumf_step_grads": (C.c_int, [_p, _pp, _pp, _p, _p, _p, _i64, _i32, _f32, _f32, _f32, _f32, _u64, _p, _p]),
    "daisy_neumf_fit_epoch": (C.c_int, [_p, _pp, _pp, _p, _p, _p, _i64, _i64, _i32, _f32, _f32, _f32, _f32, _u64, _i64,
                                        _i32, _f32, _p, _p, _p, _p, _i64, _p, _p]),
    "daisy_sgd_dense": (C.c_int, [_p, _p, _i64, _f32, _p]),
    "daisy_topk_from_scores": (C.c_int, [_p, _p, _i64, _i64, _i32, _p, _p, _sz, _p]),
    "daisy_full_topk_from_scores": (C.c_int, [_p, _i64, _i32, _p, _p, _sz, _p]),
    "daisy_gemm_nt_f32": (C.c_int, [_p, _p, _p, _i64, _i32, _i32, _p]),
    "daisy_gemm_nt_bf16": (C.c_int, [_p, _p, _p, _i64, _i32, _i32, _p]),
    "daisy_gemm_tn_bf16": (C.c_int, [_p, _p, _p, _i64, _i32, _i64, _i64, _p]),
    "daisy_gemm_nt": (C.c_int, [_p, _p, _p, _i64, _i32, _i32, _i32, _p]),
    "daisy_lgcn_graph_create": (C.c_int, [C.POINTER(_p), _p, _p, _i64, _i64, _i64, _p]),
    "daisy_lgcn_graph_destroy": (C.c_int, [_p]),
    "daisy_lgcn_graph_nnz": (_i64, [_p]),
    "daisy_lgcn_graph_set_reproducible": (C.c_int, [_p, _i32]),
    "daisy_lgcn_graph_bytes": (_sz, [_p]),
    "daisy_lgcn_graph_read": (C.c_int, [_p, _p, _p, _p, _p]),
    "daisy_lgcn_spmm": (C.c_int, [_p, _p, _p, _i32, _p]),
    "daisy_lgcn_spmm_rows": (C.c_int, [_p, _p, _p, _i32, _i64, _i64, _p]),
    "daisy_lgcn_propagate": (C.c_int, [_p, _p, _i32, _i32, _p, _p, _p]),
    "daisy_lgcn_backprop": (C.c_int, [_p, _p, _i32, _i32, _p, _p, _p]),
    "daisy_lgcn_reg_grad": (C.c_int, [_p, _p, _p, _p, _i64, _i64, _i64, _i32, _i32, _f32, _f32, _p, _p, _p, _p]),
    "daisy_axpby_f32": (C.c_int, [_p, _f32, _f32, _p, _i64, _i32, _p]),
    "daisy_csr_row_sum": (C.c_int, [_p, _p, _p, _i64, _i32, _p, _p]),
    "daisy_csr_workspace_bytes": (_sz, [_i64]),
    "daisy_build_user_csr": (C.c_int, [_p, _p, _i64, _i64, _p, _p, _p, _sz, _p]),
    "daisy_sample_neg_per_user": (C.c_int, [_p, _p, _i64, _i64, _i32, _u64, _u64, _p, _p]),
    "daisy_skipgram_samples": (C.c_int, [_p, _p, _p, _p, _i64, _i32, _p, _p, _i64, _u64, _u64, _p, _p, _p]),
    "daisy_sample_categorical": (C.c_int, [_p, _i64, _i64, _i32, _u64, _u64, _p, _i32, _i32, _p]),
    "daisy_expand_triples": (C.c_int, [_p, _p, _i64, _p, _i32, _p, _p]),
    "daisy_resample_neg_per_interaction": (C.c_int, [_p, _p, _i64, _p, _i64, _u64, _u64, _p]),
    "daisy_build_candidates": (C.c_int, [_p, _p, _p, _p, _p, _i64, _i64, _i32, _u64, _p, _p]),
    "daisy_randperm_workspace_bytes": (_sz, [_i64]),
    "daisy_randperm": (C.c_int, [_i64, _u64, _u64, _p, _p, _sz, _p]),
    "daisy_membench": (C.c_int, [_i32, _p, _i64, _i32, _p, _i64, _p, _p]),
}


class DaisyHipError(RuntimeError):
    """A HIP runtime call inside the native library failed."""


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP extension is the product path and has no "
            "fallback.  Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C daisyrec_amd/csrc` (needs hipcc, targets gfx950).")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    got = lib.daisy_abi_version()
    if got != ABI_VERSION:
        raise ImportError(f"{LIB_PATH}: ABI version {got}, binding expects {ABI_VERSION}; rebuild")
    return lib


lib = _load()


def last_error() -> str:
    msg = lib.daisy_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def check(rc: int) -> None:
    """Map the C status code to the Python exception the reference path would raise."""
    if rc == DAISY_OK:
        return
    msg = last_error()
    if rc == DAISY_ERR_ARG:
        if msg.startswith("Invalid loss type"):
            raise NotImplementedError(msg)          # MFRecommender.py:90-91
        raise ValueError(msg)
    if rc == DAISY_ERR_HIP:
        raise DaisyHipError(msg)
    raise RuntimeError(msg)
