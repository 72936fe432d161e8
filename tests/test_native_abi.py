"""The C-ABI library loads (no GPU needed) and exports exactly what include/daisyrec_amd.h
declares; the Python binding covers every declared entry point."""
import ctypes
import os
import re

from conftest import ROOT


def _declared():
    src = open(os.path.join(ROOT, "include", "daisyrec_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(daisy_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported_and_bound():
    from daisyrec_amd import _native as N
    names = _declared()
    assert len(names) >= 25
    raw = ctypes.CDLL(N.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), f"{n} declared in the header but not exported"
        assert n in N.SIGNATURES, f"{n} has no ctypes signature"
    assert sorted(N.SIGNATURES) == names
    assert N.lib.daisy_abi_version() == N.ABI_VERSION


def test_argument_errors_do_not_need_a_gpu():
    import pytest
    from daisyrec_amd import _native as N
    # NULL / out-of-range arguments are rejected before any HIP call
    rc = N.lib.daisy_bpr_ctx_create(None, 16, 64, 10, 10)
    assert rc == N.DAISY_ERR_ARG and "NULL" in N.last_error()
    h = ctypes.c_void_p()
    rc = N.lib.daisy_bpr_ctx_create(ctypes.byref(h), 16, 100000, 10, 10)
    assert rc == N.DAISY_ERR_ARG and "unsupported d" in N.last_error()
    with pytest.raises(ValueError):
        N.check(rc)
    assert N.lib.daisy_mf_rank_workspace_bytes(0, 10) == 0


def test_product_path_never_imports_oracle():
    """oracle/ is test infrastructure: nothing under daisyrec_amd/ may reference it."""
    pkg = os.path.join(ROOT, "daisyrec_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), os.path.join(dp, f)
                assert "bpr_mf_numpy" not in txt or f in ("common.h",), os.path.join(dp, f)


def test_torch_library_ops_are_registered_without_a_cpu_kernel():
    """north_star: "surfaced to Python through PyTorch-ROCm custom ops" - torch.ops.daisyrec.* exist with the
    documented schemas (in-place annotations on the tables) and have NO CPU kernel: host tensors fail in the
    dispatcher instead of reaching a fallback."""
    import pytest
    import torch
    import daisyrec_amd.torch_ops as t
    for name in t.OPS:
        assert hasattr(torch.ops.daisyrec, name)
    schema = str(torch.ops.daisyrec.bpr_mf_step.default._schema)
    assert "Tensor(a!) P" in schema and "Tensor(b!) Q" in schema
    with pytest.raises(NotImplementedError, match="CPU"):
        torch.ops.daisyrec.mf_predict(torch.zeros(3, 4), torch.zeros(3, 4), torch.zeros(2, dtype=torch.long),
                                      torch.zeros(2, dtype=torch.long))
