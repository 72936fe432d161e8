"""Property-based parity (hypothesis): random small problems - any d, any batch size, heavy id collisions
(tiny U / I so users and items repeat, items that are positive and negative at once), every loss and item
mode - one SGD step of the HIP path against the oracle."""
import numpy as np
import pytest
import torch
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

from oracle import bpr_mf_numpy as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


@st.composite
def problems(draw):
    d = draw(st.sampled_from([1, 2, 4, 7, 8, 16, 20, 32, 33, 64, 96, 128]))
    U = draw(st.integers(1, 40))
    I = draw(st.integers(1, 40))
    B = draw(st.integers(1, 700))
    seed = draw(st.integers(0, 2**31 - 1))
    loss = draw(st.sampled_from(["BPR", "HL", "TL", "CL", "SL"]))
    mode = draw(st.sampled_from(["sorted", "chunked", "atomic", "fused"]))
    reg = draw(st.sampled_from([0.0, 1e-3, 0.05]))
    return d, U, I, B, seed, loss, mode, reg


@settings(max_examples=120, deadline=None, suppress_health_check=[HealthCheck.too_slow], derandomize=True)
@given(problems())
def test_random_small_steps_match_the_oracle(p):
    from daisyrec_amd import ops
    d, U, I, B, seed, loss, mode, reg = p
    rng = np.random.default_rng(seed)
    P0 = (rng.standard_normal((U, d)) * 0.3).astype(np.float32)
    Q0 = (rng.standard_normal((I, d)) * 0.3).astype(np.float32)
    if seed % 5 == 0:
        P0[rng.integers(0, U)] = 0.0                       # an all-zero row (sign / norm subgradients)
    u = rng.integers(0, U, B).astype(np.int32)
    i = rng.integers(0, I, B).astype(np.int32)
    lt = O.LOSS_IDS[loss]
    pointwise = loss in ("CL", "SL")
    j = (rng.integers(0, 2, B) if pointwise else rng.integers(0, I, B)).astype(np.int32)
    want_loss, Pn, Qn = O.mf_sgd_step(P0, Q0, u, i, j, 0.05, reg, reg, lt)
    P, Q = torch.from_numpy(P0).to(DEV), torch.from_numpy(Q0).to(DEV)
    ctx = ops.BprContext(B, d, U, I)
    try:
        ctx.set_pointwise(pointwise)
        ctx.set_batch(*(torch.from_numpy(x).to(DEV) for x in (u, i, j)))
        sl = torch.zeros(1, dtype=torch.float64, device=DEV)
        ctx.sgd_step(P, Q, 0.05, reg, reg, loss_type=lt, item_mode=ops.ITEM_MODES[mode], step_loss=sl)
        got = float(sl.cpu())
        assert abs(got - want_loss) <= 1e-5 * abs(want_loss) + 1e-6, (p, got, want_loss)
        # fp32 sums of up to 2B terms on ONE row when U or I is 1 (the oracle sums in fp64): round-off grows
        # with the number of colliding terms; a logic error would be orders of magnitude larger
        scale = max(1.0, float(np.abs(Pn).max()), float(np.abs(Qn).max())) * max(1.0, (B / 16) ** 0.5)
        np.testing.assert_allclose(P.cpu().numpy(), Pn, rtol=0, atol=1e-5 * scale, err_msg=str(p))
        np.testing.assert_allclose(Q.cpu().numpy(), Qn, rtol=0, atol=1e-5 * scale, err_msg=str(p))
        assert float(ctx.gQ.abs().max().cpu()) == 0.0
    finally:
        ctx.close()
