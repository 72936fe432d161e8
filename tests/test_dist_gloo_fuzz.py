"""Randomised gloo runs of the user-sharded protocol (daisyrec_amd/sharding.py) on CPU ranks: random table sizes (items
fewer than ranks x slices: empty blocks and padding), world sizes 2-5, 1-6 exchange slices, the dense and the touched-rows
item exchange (round 6), SGD and the sharded Adam, steps
in which some ranks own no sample - every rank's tables and the global loss against the single-process oracle step on the
union batch (AbstractRecommender.py:119-126; the reference has no multi-device path: this is the semantics to keep).
Case k is a pure function of (DAISY_FUZZ_SEED, k); DAISY_FUZZ_CASES widens the campaign."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import bpr_mf_numpy as O

CASES = int(os.environ.get("DAISY_FUZZ_CASES", "6"))
SEED = int(os.environ.get("DAISY_FUZZ_SEED", "2022"))


def draw_case(k):
    rng = np.random.default_rng([SEED, k, 6])
    world = int(rng.integers(2, 6))
    c = dict(k=k, world=world, U=int(rng.integers(world, 200)), I=int(rng.integers(1, 120)),
             D=int(rng.choice([4, 8, 16, 20, 32])), B=int(rng.integers(1, 300)), steps=int(rng.integers(1, 5)),
             slices=int(rng.integers(1, 7)), adam=bool(rng.random() < 0.35), lopsided=bool(rng.random() < 0.5),
             overlap=bool(rng.integers(0, 2)), seed=int(rng.integers(0, 1 << 30)))
    # (round 6, a stream of its own so that the earlier draws keep their values) the item exchange: dense, the touched
    # rows only, or the automatic choice; sparse cases also draw item tables much larger than the batch
    r2 = np.random.default_rng([SEED, k, 7])
    c["exchange"] = "dense" if c["adam"] else str(r2.choice(["dense", "sparse", "sparse", "auto"]))
    if c["exchange"] != "dense" and r2.random() < 0.6:
        c["I"] = int(r2.integers(4 * c["B"] + 1, 12 * c["B"] + 50))
    return c


def _data(c):
    rng = np.random.default_rng(c["seed"])
    U, I, D, B = c["U"], c["I"], c["D"], c["B"]
    P0 = (rng.standard_normal((U, D)) * 0.2).astype(np.float32)
    Q0 = (rng.standard_normal((I, D)) * 0.2).astype(np.float32)
    batches = []
    for s in range(c["steps"]):
        nb = B if s + 1 < c["steps"] else int(rng.integers(1, B + 1))          # the epoch's last batch is partial
        b = np.stack([rng.integers(0, U, nb), rng.integers(0, I, nb), rng.integers(0, I, nb)], 1).astype(np.int32)
        if c["lopsided"] and s % 2 == 1:          # every sample belongs to the first rank's users
            b[:, 0] = rng.integers(0, max(1, U // (2 * c["world"])), nb)
        batches.append(b)
    return P0, Q0, batches


def _worker(rank, port, out_dir, c):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    world = c["world"]
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from daisyrec_amd import _native as N
    from daisyrec_amd.sharding import UserShardedBprTrainer, shard_triples, user_range
    from oracle_backend import OracleContext
    P0, Q0, batches = _data(c)
    U, I, D, B = c["U"], c["I"], c["D"], c["B"]
    lo, hi = user_range(U, world, rank)
    P = torch.from_numpy(P0[lo:hi].copy())
    Q = torch.from_numpy(Q0.copy())
    ctx = OracleContext(B, D, hi - lo, I)
    tr = UserShardedBprTrainer(ctx, P, Q, lo, 0.05 if not c["adam"] else 0.01, 0.01, 0.02, overlap=c["overlap"],
                               item_mode=N.ITEM_FUSED, slices=c["slices"], adam_steps=2 if c["adam"] else 0,
                               exchange=c["exchange"], global_batch=B)
    losses = []
    for b in batches:
        mine = shard_triples(b, U, world, rank)
        stats = tr.step_from_plan(None, 0) if len(mine) == 0 else tr.step_from_triples(torch.from_numpy(mine))
        losses.append(float(stats[7]))
    if c["adam"]:
        ctx.oracle_flush_p(P, tr.adam)
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), P=P.numpy(), Q=Q.numpy(), lo=lo, hi=hi, losses=np.array(losses),
             slices=tr.slices, sparse=int(tr.sparse))
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("k", range(CASES))
def test_random_sharded_steps_equal_the_single_process_step(tmp_path, k):
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    c = draw_case(k)
    world = c["world"]
    mp.spawn(_worker, args=(_free_port(), str(tmp_path), c), nprocs=world, join=True)
    P, Q, batches = _data(c)
    ref_losses = []
    if c["adam"]:
        ref = O.DenseAdam([P.shape, Q.shape], 0.01)
        P, Q = P.astype(np.float64), Q.astype(np.float64)
    for b in batches:
        if c["adam"]:
            loss, gP, gQ = O.mf_pair_grad(P, Q, b[:, 0], b[:, 1], b[:, 2], 0.01, 0.02)
            P, Q = ref.step([P, Q], [gP, gQ])
        else:
            loss, P, Q = O.mf_sgd_step(P, Q, b[:, 0], b[:, 1], b[:, 2], 0.05, 0.01, 0.02)
        ref_losses.append(loss)
    outs = [np.load(os.path.join(str(tmp_path), f"r{r}.npz")) for r in range(world)]
    for o in outs:
        np.testing.assert_allclose(o["losses"], ref_losses, rtol=1e-7, err_msg=str(c))       # every rank: the GLOBAL loss
        if c["adam"]:
            # (a gradient that cancels to round-off takes Adam's +-lr step with a sign the summation order decides:
            # tests/test_gpu_fuzz.py; the ranks add the item gradient in another order than one process does)
            for got, want in ((o["Q"], Q), (o["P"], P[int(o["lo"]):int(o["hi"])])):
                diff = np.abs(got - want)
                assert (diff > 2e-5).mean() < 0.02 and diff.max() <= 2.5 * 0.01 * len(batches), (c, float(diff.max()))
        else:
            np.testing.assert_allclose(o["Q"], Q, atol=2e-6, err_msg=str(c))
            np.testing.assert_allclose(o["P"], P[int(o["lo"]):int(o["hi"])], atol=2e-6, err_msg=str(c))
    if c["exchange"] == "sparse":
        assert all(int(o["sparse"]) == 1 and int(o["slices"]) == 1 for o in outs)
    for o in outs[1:]:
        np.testing.assert_array_equal(outs[0]["Q"], o["Q"], err_msg=str(c))                  # replicas stay identical
