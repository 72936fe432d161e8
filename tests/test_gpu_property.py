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


def _close(la, lb, Pa, Pb, Qa, Qb, d, nb):
    scale = max(1.0, float(np.abs(Pb).max()), float(np.abs(Qb).max()))
    tol = 3e-5 * max(1.0, d / 64) * scale * max(1, nb) ** 0.5
    return (np.allclose(la, lb, rtol=3e-5, atol=1e-6) and float(np.abs(Pa - Pb).max()) < tol
            and float(np.abs(Qa - Qb).max()) < tol and bool(np.isfinite(Pa).all()))


def test_random_small_batch_epochs_agree_with_the_phase_kernels():
    """Random epochs at B <= 256 (any d up to 256: all three staging modes of the persistent kernel, rows that do and
    do not fill their lanes, partial last batches, tiny tables so that rows recur from step to step): the
    one-workgroup epoch against the per-step phase kernels on the same plan."""
    from daisyrec_amd import ops
    rng = np.random.default_rng(1)
    for t in range(120):
        d = int(rng.choice([1, 2, 3, 4, 7, 8, 12, 16, 20, 24, 32, 33, 36, 40, 48, 50, 64, 65, 72, 96, 100, 128, 130, 160, 200, 256]))
        B = int(rng.integers(1, 257))
        U, I = int(rng.integers(1, 80)), int(rng.integers(1, 80))
        n = int(rng.integers(B, 12 * B + 1))
        loss = str(rng.choice(["BPR", "HL", "TL"]))
        reg = float(rng.choice([0.0, 1e-3, 0.05]))
        tri = np.stack([rng.integers(0, U, n), rng.integers(0, I, n), rng.integers(0, I, n)], 1).astype(np.int32)
        P0 = (rng.standard_normal((U, d)) * 0.3).astype(np.float32)
        Q0 = (rng.standard_normal((I, d)) * 0.3).astype(np.float32)
        plan = ops.EpochPlan(n, U, I).build(torch.from_numpy(tri).to(DEV), B, order="feistel", seed=t, epoch=1)
        nb = plan.num_batches
        res = []
        for mode in ("fused", "sorted"):
            P, Q = torch.from_numpy(P0).to(DEV), torch.from_numpy(Q0).to(DEV)
            ctx = ops.BprContext(B, d, U, I)
            sl = torch.zeros(nb, dtype=torch.float64, device=DEV)
            ctx.fit_epoch_sgd(plan, P, Q, 0.05, reg, 2 * reg, loss_type=ops.LOSS_IDS[loss], item_mode=ops.ITEM_MODES[mode],
                              step_losses=sl)
            torch.cuda.synchronize()
            res.append((P.cpu().numpy(), Q.cpu().numpy(), sl.cpu().numpy()))
            ctx.close()
        plan.close()
        (Pa, Qa, la), (Pb, Qb, lb) = res
        assert _close(la, lb, Pa, Pb, Qa, Qb, d, nb), dict(d=d, B=B, U=U, I=I, n=n, loss=loss, reg=reg)


def test_random_epochs_staged_step_phase_kernels_and_sliced_exchange_agree():
    """Random epochs at B > 256: the staged step over the partitioned plan, the phase kernels over the sorted plan and
    the multi-GPU form of the staged step with its item pass cut into 2..8 slices (one rank, no process group) must
    tell the same story - uniform and Zipf items, sorted and unsorted triples, d from 1 to 300."""
    from daisyrec_amd import ops
    from daisyrec_amd.sharding import UserShardedBprTrainer
    rng = np.random.default_rng(3)
    for t in range(60):
        d = int(rng.choice([1, 3, 4, 8, 16, 20, 32, 33, 48, 50, 64, 72, 96, 100, 128, 200, 256, 300]))
        B = int(rng.integers(257, 6000))
        U, I = int(rng.integers(1, 3000)), int(rng.integers(1, 2000))
        n = int(rng.integers(B, 5 * B + 1))
        loss = str(rng.choice(["BPR", "HL", "TL"]))
        reg = float(rng.choice([0.0, 1e-3, 0.05]))
        items = (rng.zipf(1.2, n) % I) if rng.random() < 0.5 else rng.integers(0, I, n)
        users = np.sort(rng.integers(0, U, n)) if rng.random() < 0.5 else rng.integers(0, U, n)
        tri = np.stack([users, items, rng.integers(0, I, n)], 1).astype(np.int32)
        P0 = (rng.standard_normal((U, d)) * 0.2).astype(np.float32)
        Q0 = (rng.standard_normal((I, d)) * 0.2).astype(np.float32)
        t_dev = torch.from_numpy(tri).to(DEV)
        lid = ops.LOSS_IDS[loss]
        index = ops.TrainIndex(t_dev, U, I)
        plan_i = ops.EpochPlan(n, U, I).build_indexed(index, B, order="feistel", seed=t, epoch=2)
        plan_s = ops.EpochPlan(n, U, I).build(t_dev, B, order="feistel", seed=t, epoch=2)
        nb = plan_i.num_batches
        res = {}
        for which in ("staged", "chunked", "slices"):
            P, Q = torch.from_numpy(P0).to(DEV), torch.from_numpy(Q0).to(DEV)
            ctx = ops.BprContext(B, d, U, I)
            sl = torch.zeros(nb, dtype=torch.float64, device=DEV)
            if which == "staged":
                ctx.fit_epoch_sgd(plan_i, P, Q, 0.05, reg, 2 * reg, loss_type=lid, item_mode=ops.ITEM_MODES["fused"], step_losses=sl)
            elif which == "chunked":
                ctx.fit_epoch_sgd(plan_s, P, Q, 0.05, reg, 2 * reg, loss_type=lid, item_mode=ops.ITEM_MODES["chunked"], step_losses=sl)
            else:
                tr = UserShardedBprTrainer(ctx, P, Q, 0, 0.05, reg, 2 * reg, loss_type=lid, slices=int(rng.integers(2, 9)))
                for k in range(nb):
                    sl[k] = tr.step_from_plan(plan_i, k)[7]
            torch.cuda.synchronize()
            res[which] = (P.cpu().numpy(), Q.cpu().numpy(), sl.cpu().numpy())
            ctx.close()
        plan_i.close(); plan_s.close(); index.close()
        Pb, Qb, lb = res["chunked"]
        for which in ("staged", "slices"):
            Pa, Qa, la = res[which]
            assert _close(la, lb, Pa, Pb, Qa, Qb, d, nb), (which, dict(d=d, B=B, U=U, I=I, n=n, loss=loss, reg=reg))
