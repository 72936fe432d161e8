"""Epoch plan (the loader boundary on the device, replaces dataset.py:5-27 iteration):
integer / index work, so every check is bit exact."""
import numpy as np
import pytest
import torch

from oracle import bpr_mf_numpy as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _triples(n, U, I, seed):
    rng = np.random.default_rng(seed)
    return np.stack([rng.integers(0, U, n), rng.integers(0, I, n), rng.integers(0, I, n)], 1).astype(np.int32)


def _check_batch(plan, k, B, want_rows, U):
    u, i, j, ei, es, eu = (t.cpu().numpy() for t in plan.read_batch(k, B))
    b = len(want_rows)
    assert len(u) == b
    # same multiset of triples as the loader's batch, grouped by user (stable => sorted by user)
    got = np.stack([u, i, j], 1)
    assert np.array_equal(got[np.lexsort((got[:, 2], got[:, 1], got[:, 0]))],
                          want_rows[np.lexsort((want_rows[:, 2], want_rows[:, 1], want_rows[:, 0]))])
    assert np.all(np.diff(u) >= 0)
    # item entries: sorted by item; each sample appears once as positive and once as negative
    assert np.all(np.diff(ei) >= 0) and len(ei) == 2 * b
    s = (es.astype(np.uint32) & np.uint32(0x7FFFFFFF)).astype(np.int64)
    neg = (es.astype(np.uint32) >> np.uint32(31)).astype(bool)
    assert np.array_equal(np.sort(s[~neg]), np.arange(b)) and np.array_equal(np.sort(s[neg]), np.arange(b))
    assert np.array_equal(ei[~neg], i[s[~neg]]) and np.array_equal(ei[neg], j[s[neg]])
    assert np.array_equal(eu, u[s])
    # within an item: positive slots first, then negative slots, each in sample order
    order_key = neg.astype(np.int64) * (1 << 40) + s
    for r in np.unique(ei)[:50]:
        m = ei == r
        assert np.all(np.diff(order_key[m]) > 0)


@pytest.mark.parametrize("n,B", [(1000, 64), (4097, 256), (513, 1000), (7, 1)])
def test_plan_orders(n, B):
    from daisyrec_amd import ops
    U, I = 300, 200
    tri = _triples(n, U, I, n)
    t_dev = torch.from_numpy(tri).to(DEV)
    plan = ops.EpochPlan(n, U, I)
    nb = (n + B - 1) // B
    # identity (shuffle=False)
    plan.build(t_dev, B, order="identity")
    assert plan.num_batches == nb
    for k in (0, nb - 1):
        _check_batch(plan, k, B, tri[k * B:(k + 1) * B], U)
    # explicit permutation (the DataLoader's RandomSampler order)
    perm = torch.randperm(n, generator=torch.Generator().manual_seed(n))
    plan.build(t_dev, B, order="perm", perm=perm.to(DEV))
    for k in range(min(nb, 3)):
        _check_batch(plan, k, B, tri[perm.numpy()[k * B:(k + 1) * B]], U)
    _check_batch(plan, nb - 1, B, tri[perm.numpy()[(nb - 1) * B:]], U)
    # device shuffle: bit-exact with the oracle's Feistel positions
    pos = ops.feistel_positions(n, 2022, 4).cpu().numpy()
    np.testing.assert_array_equal(pos, O.feistel_positions(n, 2022, 4))
    inv = np.empty(n, dtype=np.int64)
    inv[pos] = np.arange(n)                        # inv[p] = triple served at position p
    plan.build(t_dev, B, order="feistel", seed=2022, epoch=4)
    for k in (0, nb - 1):
        _check_batch(plan, k, B, tri[inv[k * B:(k + 1) * B]], U)
    plan.close()


def test_feistel_is_a_fresh_uniformish_permutation():
    from daisyrec_amd import ops
    n = 1 << 20
    p0 = ops.feistel_positions(n, 1, 0).cpu().numpy()
    p1 = ops.feistel_positions(n, 1, 1).cpu().numpy()
    assert np.array_equal(np.sort(p0), np.arange(n)) and np.array_equal(np.sort(p1), np.arange(n))
    assert (p0 == p1).mean() < 1e-4 and (p0 == np.arange(n)).mean() < 1e-4
    # positions of consecutive triples land in unrelated batches (batch ids ~ uniform)
    B = 4096
    batch = p0 // B
    cnt = np.bincount(batch[:65536], minlength=n // B)
    exp = 65536 / (n // B)
    chi2 = ((cnt - exp) ** 2 / exp).sum()
    dof = n // B - 1
    assert abs(chi2 - dof) < 6 * np.sqrt(2 * dof)
    # rank correlation between t and pos is ~0
    assert abs(np.corrcoef(np.arange(n), p0)[0, 1]) < 0.01


def test_fit_epoch_over_plan_matches_oracle_order():
    """daisy_bpr_fit_epoch_sgd over a Feistel-ordered plan == the oracle stepping through the
    same batches (batch composition is what matters; in-batch order is irrelevant to the math)."""
    from daisyrec_amd import ops
    U, I, d, n, B = 120, 90, 32, 3000, 256
    rng = np.random.default_rng(0)
    tri = _triples(n, U, I, 1)
    P0 = (rng.standard_normal((U, d)) * 0.1).astype(np.float32)
    Q0 = (rng.standard_normal((I, d)) * 0.1).astype(np.float32)
    pos = O.feistel_positions(n, 9, 0)
    inv = np.empty(n, dtype=np.int64)
    inv[pos] = np.arange(n)
    P, Q, losses = P0, Q0, []
    for s in range(0, n, B):
        rows = tri[inv[s:s + B]]
        loss, P, Q = O.mf_sgd_step(P, Q, rows[:, 0], rows[:, 1], rows[:, 2], 0.02, 1e-3, 1e-3)
        losses.append(loss)
    for mode in ("sorted", "chunked", "fused"):
        Pd, Qd = torch.from_numpy(P0).to(DEV), torch.from_numpy(Q0).to(DEV)
        ctx = ops.BprContext(B, d, U, I)
        plan = ops.EpochPlan(n, U, I).build(torch.from_numpy(tri).to(DEV), B, order="feistel", seed=9, epoch=0)
        sl = torch.zeros(plan.num_batches, dtype=torch.float64, device=DEV)
        ctx.fit_epoch_sgd(plan, Pd, Qd, 0.02, 1e-3, 1e-3, item_mode=ops.ITEM_MODES[mode], step_losses=sl)
        np.testing.assert_allclose(sl.cpu().numpy(), losses, rtol=1e-5)
        np.testing.assert_allclose(Pd.cpu().numpy(), P, atol=5e-6)
        np.testing.assert_allclose(Qd.cpu().numpy(), Q, atol=5e-6)
        assert abs(float(ctx.epoch_acc[0].cpu()) - sum(losses)) <= 1e-5 * sum(losses)
        plan.close()
        ctx.close()


def test_plan_argument_errors():
    from daisyrec_amd import ops
    plan = ops.EpochPlan(100, 10, 10)
    t = torch.zeros(100, 3, dtype=torch.int32, device=DEV)
    ctx = ops.BprContext(16, 8, 10, 10)
    with pytest.raises(RuntimeError):                    # not built yet
        ctx.set_batch_from_plan(plan, 0)
    with pytest.raises(ValueError):                      # perm order without perm
        plan.build(t, 16, order="perm")
    with pytest.raises(ValueError):                      # more triples than the plan holds
        plan.build(torch.zeros(101, 3, dtype=torch.int32, device=DEV), 16)
    plan.build(t, 16)
    with pytest.raises(ValueError):
        ctx.set_batch_from_plan(plan, 7)                 # 100/16 -> 7 batches: 0..6
    plan.build(t, 32)
    with pytest.raises(ValueError):                      # plan batch larger than the context
        ctx.set_batch_from_plan(plan, 0)
    plan.close()
    ctx.close()


def test_plan_user_sorted_fast_path_and_wide_keys():
    from daisyrec_amd import ops
    # (a) CSR-ordered triples + the one-pass partition must give the same batches as the full sort
    n, U, I, B = 5000, 400, 300, 128
    tri = _triples(n, U, I, 3)
    tri = tri[np.argsort(tri[:, 0], kind="stable")]
    t_dev = torch.from_numpy(tri).to(DEV)
    assert ops.triples_user_sorted(t_dev) and not ops.triples_user_sorted(t_dev.flip(0).contiguous())
    pos = O.feistel_positions(n, 5, 1)
    inv = np.empty(n, dtype=np.int64)
    inv[pos] = np.arange(n)
    plan = ops.EpochPlan(n, U, I)
    for fast in (True, False):
        plan.build(t_dev, B, order="feistel", seed=5, epoch=1, user_sorted=fast)
        nb = plan.num_batches
        for k in (0, 1, nb - 1):
            _check_batch(plan, k, B, tri[inv[k * B:(k + 1) * B]], U)
    plan.build(t_dev, B, order="identity", user_sorted=True)
    _check_batch(plan, 3, B, tri[3 * B:4 * B], U)
    plan.build(t_dev, n, order="identity", user_sorted=True)          # single batch: no sort at all
    _check_batch(plan, 0, n, tri, U)
    plan.close()
    # (b) batch bits + id bits > 32 -> the 64-bit key path (U = 2^20 users, 8192 one-sample batches)
    n, U, I, B = 8192, 1 << 20, 1 << 20, 1
    tri = _triples(n, U, I, 4)
    t_dev = torch.from_numpy(tri).to(DEV)
    plan = ops.EpochPlan(n, U, I).build(t_dev, B, order="identity")
    for k in (0, 17, n - 1):
        _check_batch(plan, k, B, tri[k:k + 1], U)
    B = 3
    plan.build(t_dev, B, order="feistel", seed=1, epoch=0)
    pos = O.feistel_positions(n, 1, 0)
    inv = np.empty(n, dtype=np.int64)
    inv[pos] = np.arange(n)
    for k in (0, 5, plan.num_batches - 1):
        _check_batch(plan, k, B, tri[inv[k * B:(k + 1) * B]], U)
    # a step on the wide-key plan matches the oracle
    rng = np.random.default_rng(0)
    Us, Is, d = 50, 40, 16
    tri = _triples(4096, Us, Is, 8)
    P0 = (rng.standard_normal((Us, d)) * 0.1).astype(np.float32)
    Q0 = (rng.standard_normal((Is, d)) * 0.1).astype(np.float32)
    plan.close()
