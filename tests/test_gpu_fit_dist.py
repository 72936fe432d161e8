"""`MF.fit` over the ranks of a torch.distributed job (daisyrec_amd/model/AbstractRecommender.py::_fit_sharded):
three ranks share the one GPU of the test box (gloo), each owning a user range; batch k of a rank is its share of
batch k of the single-device epoch, so the training must equal the single-process `fit` (same loader order, same
seeds) up to summation order - epoch losses, both tables, and therefore the ranked lists.  Also: the rank-share
plan itself against the whole-epoch plan (EpochPlan.build_positions vs build_indexed)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
U, I, D, N_ROWS, B, EPOCHS = 157, 211, 32, 9000, 700, 3          # 13 batches per epoch, the last one partial


SMALL = (11, 40, 16, 900, 37)      # users, items, d, rows, batch: 25 steps per epoch, ranks often without a sample in a step


def _loss():            # the loss of the current test's fits; the spawned ranks read it from the environment
    return os.environ.get("DAISY_TEST_LOSS", "BPR")


def _triples(shape=None):
    U_, I_, _, n_, _ = shape or (U, I, D, N_ROWS, B)
    rng = np.random.default_rng(5)
    u = np.sort(rng.integers(0, U_, n_))
    u[u == 3] = 4                                    # a user without interactions
    if shape is not None:
        u[: n_ // 2] = 0                             # half of the rows belong to one user (one rank)
    third = rng.integers(0, I_, n_) if _loss() in ("BPR", "HL", "TL") else rng.integers(0, 2, n_)   # negative item / label
    return np.stack([u, rng.integers(0, I_, n_), third], 1).astype(np.int32)


def _config(shuffle_mode, shape=None):
    import logging
    U_, I_, D_, _, _ = shape or (U, I, D, N_ROWS, B)
    return {"gpu": "0", "logger": logging.getLogger("t"), "lr": float(os.environ.get("DAISY_TEST_LR", "0.05")), "reg_1": 0.001, "reg_2": 0.002,
            "epochs": EPOCHS, "topk": 10, "user_num": U_, "item_num": I_, "factors": D_, "loss_type": _loss(),
            "optimizer": os.environ.get("DAISY_TEST_OPT", "sgd"), "init_method": "default", "early_stop": False, "shuffle_mode": shuffle_mode,
            "progress": False, "seed": 7}


def _fit(shuffle_mode, shuffle, shape=None):
    from daisyrec_amd.model.MFRecommender import MF
    from daisyrec_amd.model.FMRecommender import FM
    from daisyrec_amd.utils.dataset import BasicDataset, get_dataloader
    torch.manual_seed(123)
    model = (FM if os.environ.get("DAISY_TEST_MODEL") == "fm" else MF)(_config(shuffle_mode, shape))
    loader = get_dataloader(BasicDataset(_triples(shape)), batch_size=(shape or (0, 0, 0, 0, B))[4], shuffle=shuffle,
                            num_workers=0)
    torch.manual_seed(321)                           # the loader's permutations
    model.fit(loader)
    return model


def _worker(rank, world, port, out_dir, shuffle_mode, shuffle, backend="gloo", shape=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if backend == "nccl":                            # one rank per GPU over RCCL
        torch.cuda.set_device(rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    else:
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    model = _fit(shuffle_mode, shuffle, shape)
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), P=model.embed_user.weight.data.cpu().numpy(),
             Q=model.embed_item.weight.data.cpu().numpy(), losses=np.array(model.epoch_losses), **_bias_arrays(model))
    dist.destroy_process_group()


def _bias_arrays(model):
    if not hasattr(model, "u_bias"):
        return {}
    return {"bu": model.u_bias.weight.data.cpu().numpy().reshape(-1), "bi": model.i_bias.weight.data.cpu().numpy().reshape(-1),
            "b0": model.bias_.data.cpu().numpy().reshape(-1)}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _close(got, want, atol, outliers, what=""):
    """|got - want| <= atol everywhere; outliers > 0: all but that fraction of the elements (optimisers that divide by
    the root of a DECAYING state - RMSprop - turn the round-off of a near-zero gradient into a step of up to ~10 lr on
    that element, in both implementations; the bulk still agrees to atol and the rest stays within 0.02)"""
    diff = np.abs(np.asarray(got, np.float64) - np.asarray(want, np.float64))
    if outliers:
        assert (diff > atol).mean() <= outliers and diff.max() < 0.02, (what, float((diff > atol).mean()), float(diff.max()))
    else:
        np.testing.assert_allclose(got, want, atol=atol, err_msg=what)


def _compare(tmp_path, world, shuffle_mode, shuffle, shape=None, atol=5e-6, outliers=0.0):
    ref = _fit(shuffle_mode, shuffle, shape)         # no process group here: the single-device path
    P, Q = ref.embed_user.weight.data.cpu().numpy(), ref.embed_item.weight.data.cpu().numpy()
    assert len(ref.epoch_losses) == EPOCHS
    for r in range(world):
        o = np.load(os.path.join(str(tmp_path), f"r{r}.npz"))
        np.testing.assert_allclose(o["losses"], ref.epoch_losses, rtol=2e-6)
        _close(o["Q"], Q, atol, outliers, "Q")                    # (fp32 summation order differs: typically 1e-7)
        _close(o["P"], P, atol, outliers, "P")                    # every rank ends with the WHOLE user table
        for k, v in _bias_arrays(ref).items():                    # FM: every user's bias, the replicated item biases, bias_
            _close(o[k], v, atol, outliers, k)


@pytest.mark.parametrize("shuffle_mode,shuffle", [("loader", True), ("device", True), ("loader", False)])
def test_fit_over_three_ranks_equals_the_single_process_fit(tmp_path, shuffle_mode, shuffle):
    world = 3
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), shuffle_mode, shuffle), nprocs=world, join=True)
    _compare(tmp_path, world, shuffle_mode, shuffle)


@pytest.mark.parametrize("loss", ["CL", "SL"])
def test_pointwise_fit_over_three_ranks_equals_the_single_process_fit(tmp_path, loss, monkeypatch):
    """point-wise losses (rows (user, item, label), one item entry per row in the partitioned plan) through the same
    sharded protocol: the staged step covers them since round 3"""
    monkeypatch.setenv("DAISY_TEST_LOSS", loss)
    world = 3
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), "device", True), nprocs=world, join=True)
    _compare(tmp_path, world, "device", True)


@pytest.mark.parametrize("loss", ["BPR", "CL"])
def test_adam_fit_over_three_ranks_equals_the_single_process_fit(tmp_path, loss, monkeypatch):
    """torch.optim.Adam through the sharded protocol: lazy on every rank's rows of P, dense on the owner's block of Q
    after the reduce-scatter, against the single-process fit (the staged Adam step + flush per epoch).  Adam divides by
    sqrt(v): a last-bit difference in a tiny gradient moves a step by up to lr, so the tables are compared at 1e-4 of
    their scale and the losses at 2e-6."""
    monkeypatch.setenv("DAISY_TEST_LOSS", loss)
    monkeypatch.setenv("DAISY_TEST_OPT", "adam")
    world = 3
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), "device", True), nprocs=world, join=True)
    _compare(tmp_path, world, "device", True, atol=1e-4)


@pytest.mark.parametrize("loss,shape", [("BPR", None), ("CL", None), ("TL", SMALL)])
def test_fm_fit_over_three_ranks_equals_the_single_process_fit(tmp_path, loss, shape, monkeypatch):
    """FM (FMRecommender.py:61-95) sharded by user: u_bias rows travel with their users, the item-bias gradient is
    all-reduced next to the item exchange, bias_ follows from the all-reduced coefficient sum - against the
    single-process FM fit: losses, both tables, all three bias parameters.  SMALL: ranks without a sample in many steps"""
    monkeypatch.setenv("DAISY_TEST_LOSS", loss)
    monkeypatch.setenv("DAISY_TEST_MODEL", "fm")
    world = 3
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), "device", True, "gloo", shape), nprocs=world, join=True)
    _compare(tmp_path, world, "device", True, shape)


@pytest.mark.parametrize("opt,model,shape", [("adagrad", "mf", None), ("rmsprop", "mf", None), ("adam", "fm", None),
                                             ("adagrad", "fm", SMALL)])
def test_dense_optimiser_fit_over_three_ranks_equals_the_single_process_fit(tmp_path, opt, model, shape, monkeypatch):
    """torch's Adagrad / RMSprop (AbstractRecommender.py:56-61), and Adam with FM's biases, sharded by user through the
    dense-optimiser protocol (sharding.py: phase kernels, all-reduce of the dense item gradient, the dense optimiser on the
    rank's rows of P and on the replicated Q) against the single-process fit with the same optimiser.  These optimisers
    divide by the root of their state: a last-bit difference in a tiny gradient moves a step by up to lr, hence 1e-4.
    SMALL: 37-sample batches, ranks without a sample in many steps (their rows still take the optimiser's step)."""
    monkeypatch.setenv("DAISY_TEST_OPT", opt)
    monkeypatch.setenv("DAISY_TEST_MODEL", model)
    monkeypatch.setenv("DAISY_TEST_LR", {"adagrad": "0.02", "rmsprop": "0.002", "adam": "0.01"}[opt])   # (RMSprop's first steps are 10 lr)
    world = 3
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), "device", True, "gloo", shape), nprocs=world, join=True)
    _compare(tmp_path, world, "device", True, shape, atol=1e-4, outliers=5e-3 if opt == "rmsprop" else 0.0)


def test_fit_over_ranks_with_small_lopsided_batches(tmp_path):
    """37-sample global batches over 4 ranks, half of the rows owned by one rank, a rank range without any user's
    rows: most steps leave some rank without a sample (it only joins the exchanges); B = 37 <= 256 makes the
    single-device reference take the persistent small-batch kernel, the ranks the staged step"""
    world = 4
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), "loader", True, "gloo", SMALL), nprocs=world, join=True)
    _compare(tmp_path, world, "loader", True, SMALL)


def test_fit_over_ranks_lopsided_with_the_device_shuffle(tmp_path):
    """the same lopsided split with shuffle_mode='device': a rank without rows has no positions to compute (the native
    entry rejects an empty id list) and must still join every exchange - it used to raise alone and leave the other
    ranks blocked in their all-reduce"""
    world = 4
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), "device", True, "gloo", SMALL), nprocs=world, join=True)
    _compare(tmp_path, world, "device", True, SMALL)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: RCCL refuses two ranks on one device")
def test_fit_over_rccl_ranks_equals_the_single_process_fit(tmp_path):
    world = min(torch.cuda.device_count(), 8)
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), "loader", True, "nccl"), nprocs=world, join=True)
    _compare(tmp_path, world, "loader", True)


def test_rank_share_plans_tile_the_epoch_plan():
    from daisyrec_amd import ops
    from daisyrec_amd.sharding import user_range
    dev = torch.device("cuda")
    tr = torch.from_numpy(_triples()).to(dev)
    n = tr.shape[0]
    pos = ops.feistel_positions(n, 7, 2, device=dev)
    some = torch.tensor([0, 5, n - 1, 17, 5], device=dev)
    assert torch.equal(ops.feistel_positions_at(some, n, 7, 2), pos[some])
    assert ops.feistel_positions_at(some[:0], n, 7, 2).numel() == 0
    whole = ops.EpochPlan(n, U, I, device=dev).build_indexed(ops.TrainIndex(tr, U, I), B, order="feistel", seed=7, epoch=2)
    nb = whole.num_batches
    got = [[] for _ in range(nb)]
    world = 4
    for r in range(world):
        lo, hi = user_range(U, world, r)
        ids = torch.nonzero((tr[:, 0] >= lo) & (tr[:, 0] < hi)).flatten()
        mine = tr[ids].contiguous()
        index = ops.TrainIndex(mine, hi - lo, I, user_base=lo)
        plan = ops.EpochPlan(mine.shape[0], hi - lo, I, device=dev).build_positions(index, pos[ids].contiguous(), n, B)
        assert plan.num_batches == nb
        assert sum(plan.batch_rows(k) for k in range(nb)) == mine.shape[0]
        for k in range(nb):
            if plan.batch_rows(k) == 0:
                continue
            u, i, j, ei, es, _ = plan.read_batch(k, B)
            assert u.shape[0] == plan.batch_rows(k)
            got[k].append(torch.stack([u + lo, i, j], 1).cpu().numpy())
            # entries: sorted by item, their stage slots are the positions of the samples inside the GLOBAL batch
            assert bool((ei[1:] >= ei[:-1]).all())
            slots = (es & 0x7FFFFFFF).cpu().numpy()
            assert slots.min() >= 0 and slots.max() < B
        plan.close(); index.close()
    for k in range(nb):
        u, i, j, *_ = whole.read_batch(k, B)
        want = torch.stack([u, i, j], 1).cpu().numpy()
        have = np.concatenate(got[k])
        assert have.shape == want.shape
        key = lambda a: a[np.lexsort((a[:, 2], a[:, 1], a[:, 0]))]
        np.testing.assert_array_equal(key(have), key(want))        # the ranks' batches k are a partition of batch k
    with pytest.raises(ValueError, match="outside"):
        bad = pos.clone(); bad[0] = n + 5
        ops.EpochPlan(n, U, I, device=dev).build_positions(ops.TrainIndex(tr, U, I), bad, n, B)


def test_rank_shares_at_baseline_scale_tile_the_epoch():
    """BASELINE configs[1] shapes (1 M users x 100 K items, 2 M-sample batches) cut for 8 ranks: size-independent
    properties of the rank-share plans - every row lands in exactly one (rank, batch), the batches of the ranks add
    up to the global batch sizes, every batch is grouped by user and its entries sorted by item, stage slots are
    distinct positions inside the global batch."""
    import bench
    from daisyrec_amd import ops
    from daisyrec_amd.sharding import user_range
    dev = torch.device("cuda")
    U_, I_, n_, B_ = 1_000_000, 100_000, 20_000_000, 1 << 21
    tr = bench.synth_triples(U_, I_, n_, 2022, dev)
    n = tr.shape[0]
    pos = ops.feistel_positions(n, 5, 1, device=dev)
    nb = (n + B_ - 1) // B_
    total = torch.zeros(nb, dtype=torch.int64)
    seen = torch.zeros(n, dtype=torch.int32, device=dev)
    world = 8
    for r in range(world):
        lo, hi = user_range(U_, world, r)
        ids = torch.nonzero((tr[:, 0] >= lo) & (tr[:, 0] < hi)).flatten()
        mine = tr[ids].contiguous()
        index = ops.TrainIndex(mine, hi - lo, I_, user_base=lo, user_sorted=True)
        plan = ops.EpochPlan(mine.shape[0], hi - lo, I_, device=dev).build_positions(index, pos[ids].contiguous(), n, B_)
        assert plan.num_batches == nb
        rows = torch.tensor([plan.batch_rows(k) for k in range(nb)])
        assert int(rows.sum()) == mine.shape[0]
        total += rows
        for k in (0, nb // 2, nb - 1):
            u, i, j, ei, es, _ = plan.read_batch(k, B_)
            assert bool((u[1:] >= u[:-1]).all()) and bool((ei[1:] >= ei[:-1]).all())
            slots = (es & 0x7FFFFFFF).to(torch.int64)
            assert int(slots.min()) >= 0 and int(slots.max()) < B_
            pos_slots = slots[::1].unique()
            assert pos_slots.numel() == u.shape[0]           # two entries per sample, one slot per sample
            seen.index_add_(0, (pos_slots + k * B_).clamp_(max=n - 1), torch.ones_like(pos_slots, dtype=torch.int32))
        plan.close(); index.close()
    want = torch.full((nb,), B_, dtype=torch.int64)
    want[-1] = n - (nb - 1) * B_
    assert torch.equal(total, want)                          # the ranks' batches k add up to the global batch k
    assert int(seen.max()) <= 1                              # no epoch position claimed twice
