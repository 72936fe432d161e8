"""torch.ops.daisyrec.* (csrc/torch_ops.cpp) against the oracle: the shim calls the same entry points as the
ctypes binding."""
import numpy as np
import pytest
import torch

from oracle import bpr_mf_numpy as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_torch_ops_match_the_oracle():
    import daisyrec_amd.torch_ops  # noqa: F401
    rng = np.random.default_rng(3)
    U, I, d, B = 120, 90, 32, 700
    P0 = (rng.standard_normal((U, d)) * 0.1).astype(np.float32)
    Q0 = (rng.standard_normal((I, d)) * 0.1).astype(np.float32)
    u, i, j = (rng.integers(0, n, B).astype(np.int32) for n in (U, I, I))
    P, Q = torch.from_numpy(P0).to(DEV), torch.from_numpy(Q0).to(DEV)
    td = lambda x: torch.from_numpy(x).to(DEV)
    Pn, Qn = P0, Q0
    for _ in range(3):       # P changes between calls: the op must not trust a cache across calls
        loss = torch.ops.daisyrec.bpr_mf_step(P, Q, td(u), td(i), td(j), 0.05, 1e-3, 2e-3, 1e-10, 0)
        want, Pn, Qn = O.mf_sgd_step(Pn, Qn, u, i, j, 0.05, 1e-3, 2e-3)
        assert abs(float(loss.cpu()) - want) <= 1e-5 * abs(want)
        P.mul_(1.0)
    assert np.abs(P.cpu().numpy() - Pn).max() < 5e-6 and np.abs(Q.cpu().numpy() - Qn).max() < 5e-6
    with pytest.raises(RuntimeError, match="out of range"):
        bad = j.copy()
        bad[5] = I
        torch.ops.daisyrec.bpr_mf_step(P, Q, td(u), td(i), td(bad), 0.05, 0.0, 0.0, 1e-10, 0)
    us = torch.arange(10, device=DEV)
    cands = td(rng.integers(0, I, (10, 40)))
    ids = torch.ops.daisyrec.mf_rank_topk(P, Q, us, cands, 7)
    want, _ = O.mf_rank(P.cpu().numpy(), Q.cpu().numpy(), us.cpu().numpy(), cands.cpu().numpy(), 7)
    np.testing.assert_array_equal(ids.cpu().numpy().astype(np.float32), want)
    np.testing.assert_array_equal(torch.ops.daisyrec.mf_full_rank(P, Q, 4, 9).cpu().numpy(),
                                  O.mf_full_rank(P.cpu().numpy(), Q.cpu().numpy(), 4, 9))
    pr = torch.ops.daisyrec.mf_predict(P, Q, us, us)
    np.testing.assert_allclose(pr.cpu().numpy(), O.mf_forward(P.cpu().numpy(), Q.cpu().numpy(), np.arange(10), np.arange(10)),
                               atol=1e-6)
    from daisyrec_amd import ops
    users = td(np.repeat(np.arange(U, dtype=np.int32), 3))
    items = td(np.tile(np.arange(3, dtype=np.int32), U))
    indptr, csr = ops.build_user_csr(users, items, U)
    js = torch.ops.daisyrec.sample_uniform_neg(indptr, csr, I, 4, 11, 0)
    np.testing.assert_array_equal(js.cpu().numpy(), ops.sample_neg_per_user(indptr, csr, I, 4, 11, 0).cpu().numpy())
    assert int(js.min()) >= 3
