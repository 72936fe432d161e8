"""Host-side mirror of the reference interface (no GPU): construction, init parity,
DataLoader order replay, error behaviour."""
import numpy as np
import pytest
import torch

from conftest import mf_config


def test_mf_init_matches_reference(ml100k):
    """MFRecommender.py:53-61: tables are normal(0,0.01) from the global torch RNG, user first."""
    from daisyrec_amd.model.MFRecommender import MF
    g = ml100k
    torch.manual_seed(int(g["seed"]))
    m = MF(mf_config(user_num=int(g["user_num"]), item_num=int(g["item_num"])))
    np.testing.assert_array_equal(m.embed_user.weight.detach().numpy(), g["P0"])
    np.testing.assert_array_equal(m.embed_item.weight.detach().numpy(), g["Q0"])
    assert m.optimizer == "sgd" and m.initializer == "normal" and m.loss_type == "BPR"
    assert set(m.state_dict()) == {"embed_user.weight", "embed_item.weight"}


@pytest.mark.parametrize("num_workers", [0, 2])
def test_epoch_order_replays_dataloader(num_workers):
    """GeneralRecommender._epoch_order consumes the global RNG exactly like iterating the
    reference's DataLoader (dataset.py:5-7), for every epoch."""
    from daisyrec_amd.model.AbstractRecommender import GeneralRecommender
    from daisyrec_amd.utils.dataset import BasicDataset, get_dataloader
    n, B = 1000, 64
    data = np.stack([np.arange(n), np.arange(n) + 1, np.arange(n) + 2], 1).astype(np.int32)
    torch.manual_seed(123)
    loader = get_dataloader(BasicDataset(data), batch_size=B, shuffle=True, num_workers=num_workers)
    want = []
    for _ in range(3):
        want.append(torch.cat([b[0] for b in loader]).numpy())
    torch.manual_seed(123)
    loader2 = get_dataloader(BasicDataset(data), batch_size=B, shuffle=True, num_workers=0)
    for ep in range(3):
        perm = GeneralRecommender._epoch_order(loader2, n)
        np.testing.assert_array_equal(data[perm.numpy(), 0], want[ep])
    seq = get_dataloader(BasicDataset(data), batch_size=B, shuffle=False, num_workers=0)
    assert GeneralRecommender._epoch_order(seq, n) is None


def test_error_behaviour_matches_reference():
    from daisyrec_amd.model.MFRecommender import MF
    m = MF(mf_config(user_num=5, item_num=7, loss_type="XX"))
    with pytest.raises(NotImplementedError):          # AbstractRecommender.py:91
        m._build_criterion(m.loss_type)
    m = MF(mf_config(user_num=5, item_num=7, optimizer="nonsense"))
    assert m._resolve_optimizer() == "adam"           # AbstractRecommender.py:63-65
    for name in ("rmsprop", "Adagrad", "SGD", "adam"):         # AbstractRecommender.py:52-61 (case-insensitive)
        assert MF(mf_config(user_num=5, item_num=7, optimizer=name))._resolve_optimizer() == name.lower()
    m = MF(mf_config(user_num=5, item_num=7, optimizer="sparse_adam"))
    with pytest.raises(RuntimeError, match="SparseAdam does not support dense gradients"):
        m._resolve_optimizer()                        # what optim.SparseAdam.step() says on the reference's dense grads
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            m.fit(None)


def test_ops_reject_host_tensors():
    from daisyrec_amd import ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.mf_predict(torch.zeros(4, 8), torch.zeros(4, 8), torch.zeros(2, dtype=torch.int64),
                       torch.zeros(2, dtype=torch.int64))
    with pytest.raises(NotImplementedError):
        ops.loss_id("CLX")


def test_get_ur_get_ir_build_the_reference_dicts(ml100k):
    """utils.py:19-51 without the row loop: same keys, same sets, same container type; on the ml-100k train set and
    on edge cases (empty frame, duplicate rows, one user).  Where the reference checkout is reachable, its own
    functions are the judge; everywhere, a literal restatement of its loop."""
    import importlib
    import os
    import sys
    from collections import defaultdict
    import pandas as pd
    from daisyrec_amd.utils.utils import get_ir, get_ur

    def loop(df, a, b):                       # utils.py:29-32 / 46-49 as written
        out = defaultdict(set)
        for _, row in df.iterrows():
            out[int(row[a])].add(int(row[b]))
        return out

    g = ml100k
    big = pd.DataFrame({"user": g["train_users"][:20000], "item": g["train_items"][:20000], "rating": 1.0})
    frames = [big, big.iloc[:0], pd.DataFrame({"user": [3, 3, 3], "item": [7, 7, 1], "rating": 1.0}),
              pd.DataFrame({"user": [5], "item": [0], "rating": 1.0})]
    for df in frames:
        ur, ir = get_ur(df), get_ir(df)
        assert type(ur) is defaultdict and ur == loop(df, "user", "item") and ir == loop(df, "item", "user")
        assert all(type(k) is int for k in ur) and all(type(i) is int for s in ur.values() for i in s)
    ref = os.environ.get("DAISY_REFERENCE", "/root/reference")
    if os.path.isdir(os.path.join(ref, "daisy")):
        sys.path.insert(0, ref)
        sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden", "_shims"))
        try:
            U = importlib.import_module("daisy.utils.utils")
            assert get_ur(big) == U.get_ur(big) and get_ir(big) == U.get_ir(big)
        finally:
            sys.path.remove(ref)


def test_bench_byte_model_and_roofline_arithmetic():
    """bench.py's roofline object (task brief 4): SURVEY 8(d)'s 24 d + 12 B per interaction, priced against 8 TB/s; at
    N = 1 the achieved rate is never better than the wall clock says (a GPU-event mean below it is ignored)"""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    assert bench.ALGO_BYTES_PER_INTERACTION_SGD(64) == 1548 and bench.ALGO_BYTES_PER_INTERACTION_SGD(32) == 780
    assert bench.HBM_PEAK_GBS == 8000.0
    B, steps = 1 << 21, 10
    r = {"step_ms": [0.60] * steps, "dt": 0.0066, "steps": steps, "d": 64, "B": B}          # wall: 0.66 ms per step
    ach, gpu_ms = bench.roofline_of(r, 1)
    assert abs(gpu_ms - 0.60) < 1e-12
    assert abs(ach - 1548 * B / 0.66e-3 / 1e9) < 1e-6 * ach                                # the slower of the two clocks
    r["dt"] = 0.0050                                                                        # wall below the event mean
    ach, _ = bench.roofline_of(r, 1)
    assert abs(ach - 1548 * B / 0.60e-3 / 1e9) < 1e-6 * ach
    ach8, _ = bench.roofline_of(dict(r, dt=0.0090), 8)                                      # N > 1: per-GPU event time
    assert abs(ach8 - 1548 * B / 0.60e-3 / 1e9) < 1e-6 * ach8
    assert 0.0 < ach / bench.HBM_PEAK_GBS < 1.0


def test_bench_cpu_leg_runs_and_reports_its_three_samples():
    """bench.py's cpu_baseline (the only place outside tests/ and smoke() that may touch oracle/): the reference's op
    composition on the host at the headline batch, the >= 30 steps of SURVEY 8(d) at a smaller batch of the same epoch,
    and the reference's default batch - on a miniature here: the keys the driver's line carries, positive rates, the
    step counts clamped to what the batch holds."""
    import torch
    import bench
    g = torch.Generator().manual_seed(3)
    U, I, d, B = 500, 300, 16, 8192
    batches = [tuple(torch.randint(0, hi, (B,), generator=g) for hi in (U, I, I)) for _ in range(2)]
    out = bench.cpu_baseline(U, I, d, B, batches, 1e-3)
    assert out["kind"] == "port" and out["unit"] == "interactions/s" and out["value"] > 0 and out["cores"] >= 1
    assert "2 SGD steps at B=8192" in out["sample"]
    mid, small = out["steps30_b65536"], out["reference_default_batch"]
    assert mid["batch"] == 8192 and mid["steps"] == 1 and mid["value"] > 0          # (a batch of 8192 rows holds one slice of its size)
    assert small["batch"] == 256 and small["steps"] == 10 and small["value"] > 0


def test_auto_exchange_slices_properties():
    """sharding.auto_exchange_slices (`slices='auto'`): a power of two up to 16, one without RCCL or without peers, never
    a block under 1024 rows per owner and slice nor a slice under ~100 us of item pass, one slice where the exchange is
    under a tenth of the pass, and a slower bus never asks for fewer"""
    from daisyrec_amd.sharding import auto_exchange_slices as f
    rng = np.random.default_rng(4)
    for _ in range(400):
        I = int(np.exp(rng.uniform(np.log(10), np.log(5e6))))
        d = int(rng.choice([8, 32, 64, 128, 256]))
        world = int(rng.integers(1, 17))
        B = int(np.exp(rng.uniform(np.log(1), np.log(3e7))))
        s = f(I, d, world, B)
        assert s in (1, 2, 4, 8, 16)
        assert f(I, d, world, B, backend="gloo") == 1 and f(I, d, 1, B) == 1
        if s > 1:
            assert I // (world * s) >= 1024 and 2.0 * B * 4.0 * d / 5.0e12 / s >= 100e-6
        t_ex = 2.0 * (world - 1) / world * I * (d + 2) * 4.0 / 300e9
        if t_ex < 0.1 * (2.0 * B * 4.0 * d / 5.0e12):
            assert s == 1
        assert f(I, d, world, B, bus_gbs=50.0) >= f(I, d, world, B, bus_gbs=2000.0)      # a slower bus never asks for fewer
