"""Randomised differential test of the staged epoch (csrc/bpr_staged.hip) at the sizes between the unit cases and the
bench: batches of 1 ... 200 000 samples, tables of 1 ... 80 000 rows, any factor count, uniform and Zipf ids (hot users
and items: long runs, long segments, long edge chains), every loss, SGD and torch's Adam, the item pass's flavours and
launch forms forced on and off - one epoch (up to ~3 batches, the last one partial) through the partitioned plan and
daisy_bpr_fit_epoch_sgd / daisy_bpr_fit_epoch_adam against the oracle (MFRecommender.py:63-97,
AbstractRecommender.py:119-126) on the batches the plan serves.

Case k is a pure function of (DAISY_FUZZ_SEED, k): a failure names its case and reproduces alone.  DAISY_FUZZ_CASES
(default 36) widens the campaign: the round's long run is recorded in profiles/r05_fuzz.txt."""
import os

import numpy as np
import pytest
import torch

from oracle import bpr_mf_numpy as O

pytestmark = pytest.mark.gpu
DEV = "cuda"
CASES = int(os.environ.get("DAISY_FUZZ_CASES", "36"))
SEED = int(os.environ.get("DAISY_FUZZ_SEED", "2022"))


def _log_uniform(rng, lo, hi):
    return int(round(float(np.exp(rng.uniform(np.log(lo), np.log(hi))))))


def _ids(rng, n, size, alpha):
    """n ids in [0, size): uniform (alpha = 0) or Zipf(alpha) over a random ranking of the rows"""
    if alpha == 0.0 or size == 1:
        return rng.integers(0, size, n)
    w = 1.0 / np.arange(1, size + 1) ** alpha
    return rng.permutation(size)[rng.choice(size, n, p=w / w.sum())]


def draw_case(k):
    rng = np.random.default_rng([SEED, k])
    d = int(rng.choice([4, 8, 16, 20, 32, 50, 64, 64, 64, 100, 128, 200, 256]))
    B = _log_uniform(rng, 1, 200_000)
    B = max(1, min(B, 8_000_000 // d))                       # (the oracle's share of the run time)
    U = _log_uniform(rng, 1, 80_000)
    I = _log_uniform(rng, 1, 40_000)
    n = max(1, min(int(B * rng.uniform(1.0, 3.3)), 400_000))
    loss = str(rng.choice(["BPR", "BPR", "BPR", "HL", "TL", "CL", "SL"]))
    opt = "adam" if rng.random() < 0.3 else "sgd"
    a_u = float(rng.choice([0.0, 0.0, 0.7, 1.0, 1.3]))
    a_i = float(rng.choice([0.0, 0.0, 0.7, 1.0, 1.3]))
    reg = [(0.0, 0.0), (1e-3, 2e-3), (0.01, 0.0), (0.0, 5e-3)][int(rng.integers(0, 4))]
    env = {"DAISY_STAGED_SPARSE": rng.choice([None, None, "0", "1"]), "DAISY_STAGED_MERGE": rng.choice([None, None, "0", "1"]),
           "DAISY_EDGE_BLOCKS": rng.choice([None, None, "0", "1"])}
    return dict(k=k, d=d, B=B, U=U, I=I, n=n, loss=loss, opt=opt, a_u=a_u, a_i=a_i, reg=reg,
                env={key: (None if v is None else str(v)) for key, v in env.items()}, rng=rng)


@pytest.mark.parametrize("k", range(CASES))
def test_random_epoch_matches_the_oracle(k, monkeypatch):
    from daisyrec_amd import ops
    c = draw_case(k)
    rng, d, B, U, I, n = c["rng"], c["d"], c["B"], c["U"], c["I"], c["n"]
    tag = {key: v for key, v in c.items() if key != "rng"}
    for key, v in c["env"].items():
        if v is not None:
            monkeypatch.setenv(key, v)
    point = c["loss"] in ("CL", "SL")
    if not point and I == 1:
        I = 2                              # (a pair needs two items)
        tag["I"] = 2
    pos = _ids(rng, n, I, c["a_i"])
    # a negative is never the sample's positive (sampler.py:82-89 draws from the complement of the user's items)
    third = rng.integers(0, 2, n) if point else (pos + rng.integers(1, I, n)) % I
    tri = np.stack([_ids(rng, n, U, c["a_u"]), pos, third], 1).astype(np.int32)
    P0 = (rng.standard_normal((U, d)) * 0.1).astype(np.float32)
    Q0 = (rng.standard_normal((I, d)) * 0.1).astype(np.float32)
    # the loss is a SUM over the batch (loss.py:11): a row with m samples in a batch moves m times as far.  The rate keeps
    # the hottest row's step of the order of its size, as a user of the reference would have to
    hot = max(int(np.bincount(tri[:, 0]).max()), int(np.bincount(tri[:, 1]).max()))
    lr = 0.01 if c["opt"] == "adam" else 0.05 / max(1.0, hot / 50.0)
    if c["loss"] == "SL" and c["opt"] == "sgd":
        lr /= max(1.0, d / 32.0)           # (MSE: curvature 2 |row|^2 = 0.02 d per sample - keep the hot rows' steps stable)
    reg_1, reg_2 = c["reg"]
    lid = ops.LOSS_IDS[c["loss"]]

    t_dev = torch.from_numpy(tri).to(DEV)
    index, plan = ops.TrainIndex(t_dev, U, I, pointwise=point), ops.EpochPlan(n, U, I)
    plan.build_indexed(index, B, order="feistel", seed=SEED % 1000, epoch=k)
    nb = plan.num_batches
    assert nb == (n + B - 1) // B
    P, Q = torch.from_numpy(P0).to(DEV), torch.from_numpy(Q0).to(DEV)
    ctx = ops.BprContext(B, d, U, I)
    sl = torch.zeros(nb, dtype=torch.float64, device=DEV)
    if c["opt"] == "sgd":
        ctx.fit_epoch_sgd(plan, P, Q, lr, reg_1, reg_2, loss_type=lid, item_mode=ops.ITEM_MODES["fused"], step_losses=sl)
    else:
        adam = ops.LazyAdam(P, Q, lr, nb)
        adam.fit_epoch(ctx, plan, reg_1, reg_2, lid, step_losses=sl)
    torch.cuda.synchronize()
    assert float(ctx.epoch_acc[1].cpu()) == 0.0, tag
    got_l, got_P, got_Q = sl.cpu().numpy(), P.cpu().numpy(), Q.cpu().numpy()

    Pn, Qn = P0.astype(np.float64), Q0.astype(np.float64)
    ref = O.DenseAdam([P0.shape, Q0.shape], lr) if c["opt"] == "adam" else None
    cnt_u, cnt_i = np.zeros(U), np.zeros(I)
    step_u, step_i = np.zeros(U), np.zeros(I)               # the largest step a row took (its largest element)
    frail = [np.zeros(P0.shape, bool), np.zeros(Q0.shape, bool)]
    forked = fork_next = False
    seen = 0
    for b in range(nb):
        u, i, j = (t.cpu().numpy().astype(np.int64) for t in plan.read_batch(b, B)[:3])
        seen += len(u)
        if ref is None:
            Pb, Qb = Pn, Qn
            want, Pn, Qn = O.mf_sgd_step(Pn, Qn, u, i, j, lr, reg_1, reg_2, loss_type=lid)
            step_u = np.maximum(step_u, np.abs(Pn - Pb).max(1))
            step_i = np.maximum(step_i, np.abs(Qn - Qb).max(1))
        else:
            if point:
                want, gP, gQ = O.mf_point_grad(Pn, Qn, u, i, j, reg_1, reg_2, lid)
            else:
                want, gP, gQ = O.mf_pair_grad(Pn, Qn, u, i, j, reg_1, reg_2, lid)
            # elements of TOUCHED rows whose gradient all but cancels: Adam steps by lr * m / sqrt(v) ~ +-lr whatever the
            # size of g, so the round-off of an fp32 sum of `cnt` terms (`noise`: ~6e-8 sqrt(cnt) of the terms' sizes, <= 0.3
            # each) decides a visible share of such a step - in torch's fp32 as much as here.  Frail: the round-off may
            # reach a thousandth of the element - not compared.  Where it may reach the element's own size the step's SIGN
            # is open, and what meets that row in a later step follows another valid trajectory (`forked`)
            forked = forked or fork_next
            for t, g, rows in ((0, gP, u), (1, gQ, i if point else np.concatenate([i, j]))):
                cnt_b = np.bincount(rows, minlength=g.shape[0]).astype(np.float64)
                noise = (6e-8 * np.sqrt(cnt_b) * 0.3 * cnt_b)[:, None]
                frail[t] |= (cnt_b > 0)[:, None] & (np.abs(g) < 1e3 * noise + 1e-4 * np.abs(g).max())
                fork_next = fork_next or bool(((cnt_b > 0)[:, None] & (np.abs(g) <= 3.0 * noise)).any())
            Pn, Qn = ref.step([Pn, Qn], [gP, gQ])
        # (Adam: once a frail element - above - has taken its +-lr step by round-off, the trajectories are two valid ones)
        rel = 2e-5 if not forked else 2e-3
        assert abs(got_l[b] - want) <= rel * abs(want) + 1e-6, (tag, b, got_l[b], want)
        cnt_u = np.maximum(cnt_u, np.bincount(u, minlength=U))
        ci = np.bincount(i, minlength=I)
        cnt_i = np.maximum(cnt_i, ci if point else ci + np.bincount(j, minlength=I))
    assert seen == n, tag                                   # the plan served every row exactly once (sizes; ids: test_gpu_plan)
    worst, n_forked = 0.0, 0
    for name, got, want_t, cnt, step, fr in (("P", got_P, Pn, cnt_u, step_u, frail[0]), ("Q", got_Q, Qn, cnt_i, step_i, frail[1])):
        diff = np.abs(got - want_t)
        if ref is None:
            # a row's step is lr times an fp32 sum of `cnt` terms of size <= ~0.3 (coefficient x row, the regulariser's share)
            # where the oracle sums in fp64: round-off ~ sqrt(cnt) ulps of the sum of the terms' sizes (a user's edge
            # chain is added link by link: 20 000 samples of ONE user in a batch measure 4e-6 of the step, where the item
            # pass's two-level chains stay at 1e-7 - profiles/r05_fuzz.txt), the coefficients' own ~1e-6 relative error, and the stored
            # row's rounding once per step.  A missing or doubled term is lr * 0.05 or more
            tol = (3e-7 + lr * 0.3 * (5e-8 * cnt ** 1.5 + 4e-7 * cnt) + 8e-7 * np.sqrt(cnt) * step)[:, None]
            bad = diff > tol
            worst = max(worst, float((diff / tol).max()))
            assert not bad.any(), (tag, name, int(bad.sum()), float(diff.max()), float((diff / tol).max()),
                                   np.argwhere(bad)[:4].tolist())
        else:
            # Adam divides by sqrt(v): an element whose gradient is ~0 amplifies fp32 round-off to a visible fraction of
            # lr (tests/test_gpu_staged.py::test_staged_adam_epochs_match_the_dense_oracle); a logic error moves whole rows
            # every element moves by ~lr per step at most: two valid trajectories are never further apart than this
            assert diff.max() <= 2.5 * lr * nb, (tag, name, float(diff.max()))
            if not forked:
                diff = np.where(fr, 0.0, diff)
                assert (diff > 2e-5).sum() <= max(2, 2e-3 * diff.size), (tag, name, int((diff > 2e-5).sum()), float(diff.max()))
                assert diff.shape[0] < 8 or np.median(diff) < 1e-7, (tag, name, float(np.median(diff)))
            else:
                n_forked += 1
    if os.environ.get("DAISY_FUZZ_LOG"):
        with open(os.environ["DAISY_FUZZ_LOG"], "a") as f:
            f.write(f"{tag} lr={lr:.3g} hot={hot} nb={nb} worst diff/tol={worst:.3f} forked={n_forked > 0}\n")
    ctx.close(); plan.close(); index.close()
