"""LightGCN over several ranks (BASELINE configs[4]): the row-range product equals the rows of the full
product, and a whole training with the propagation row-sharded over 3 ranks (gloo; the ranks share the test
box's one GPU, RCCL needs one GPU per rank) reproduces the single-process training.  The RCCL variant runs on
the first box with >= 2 GPUs."""
import os
import socket

import numpy as np
import pytest
import scipy.sparse as sp
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import mf_config
from oracle import lightgcn_numpy as LG

pytestmark = pytest.mark.gpu
DEV = "cuda"
U, I, D, NNZ, L = 700, 450, 64, 9000, 3


def _graph_data():
    rng = np.random.default_rng(5)
    gu, gi = rng.integers(0, U, NNZ), rng.integers(0, I, NNZ)
    gi[:1500] = 3                                             # a very popular item: a row spanning many chunks
    samples = np.stack([gu, gi, rng.integers(0, I, NNZ)], 1).astype(np.int32)[:4096]
    return gu, gi, samples


@pytest.mark.parametrize("reproducible", [False, True])
@pytest.mark.parametrize("d", [64, 20])
def test_row_range_products_tile_the_full_product(reproducible, d):
    from daisyrec_amd import ops
    gu, gi, _ = _graph_data()
    g = ops.LgcnGraph(torch.from_numpy(gu).to(DEV), torch.from_numpy(gi).to(DEV), U, I)
    g.set_reproducible(reproducible)
    N = U + I
    X = torch.randn(N, d, device=DEV)
    full = g.spmm(X).cpu().numpy()
    want = LG.spmm(LG.norm_adj_csr(gu, gi, U, I), X.cpu().numpy().astype(np.float64))
    assert np.abs(full - want).max() < 1e-5
    for world in (1, 3, 8):
        rows = (N + world - 1) // world
        for r in range(world):
            lo, hi = min(r * rows, N), min((r + 1) * rows, N)
            blk = torch.full((rows + 2, d), 7.0, device=DEV)
            own = g.spmm_rows(X, blk, lo, hi).cpu().numpy()
            if reproducible:
                assert np.array_equal(own, full[lo:hi]), (world, r)
            else:                     # chunk boundaries fall elsewhere: same sums, another order
                assert np.abs(own - full[lo:hi]).max() < 2e-6, (world, r)
    g.close()


@pytest.mark.parametrize("d", [64, 20])
def test_row_range_products_with_isolated_nodes_at_the_boundaries(d):
    """user_num / item_num come from the full dataset, the graph from the train split: nodes without an edge are
    common.  An odd entry range next to such a node must not spill into a row that is not adjacent (the
    borrowed entry's partial sum used to land outside the caller's block); the guard rows around every block stay
    untouched beyond the two spare rows."""
    from daisyrec_amd import ops
    rng = np.random.default_rng(11)
    Uq, Iq, n = 96, 64, 1500
    N = Uq + Iq
    for world in (2, 3, 5, 8):
        rows = (N + world - 1) // world
        gu, gi = rng.integers(0, Uq, n), rng.integers(0, Iq, n)
        # empty the nodes on either side of every shard boundary (and one whole shard's first rows)
        dead = set()
        for r in range(1, world):
            dead.update({r * rows - 1, r * rows, r * rows + 1})
        keep = np.array([(u not in dead) and ((Uq + i) not in dead) for u, i in zip(gu, gi)])
        gu, gi = gu[keep], gi[keep]
        if len(gu) % 2 == 0:                                      # make odd ranges likely on both sides
            gu, gi = gu[:-1], gi[:-1]
        g = ops.LgcnGraph(torch.from_numpy(gu).to(DEV), torch.from_numpy(gi).to(DEV), Uq, Iq)
        X = torch.randn(N, d, device=DEV)
        want = LG.spmm(LG.norm_adj_csr(gu, gi, Uq, Iq), X.cpu().numpy().astype(np.float64))
        for r in range(world):
            lo, hi = min(r * rows, N), min((r + 1) * rows, N)
            guard = 4
            big = torch.full((rows + 2 + 2 * guard, d), 7.0, device=DEV)
            blk = big[guard:guard + rows + 2]
            own = g.spmm_rows(X, blk, lo, hi).cpu().numpy()
            assert np.abs(own - want[lo:hi]).max() < 2e-6, (world, r)
            assert bool((big[:guard] == 7.0).all()) and bool((big[guard + rows + 2:] == 7.0).all()), (world, r)
        g.close()


def _train(shard):
    from daisyrec_amd.model.LightGCNRecommender import LightGCN
    from daisyrec_amd.utils.dataset import BasicDataset, get_dataloader
    gu, gi, samples = _graph_data()
    torch.manual_seed(0)
    cfg = mf_config(user_num=U, item_num=I, factors=D, num_layers=L, algo_name="lightgcn", reg_1=1e-3, reg_2=1e-3,
                    lr=0.01, epochs=2, batch_size=512, shard_rows=shard, item_mode="sorted",
                    propagation_pieces=int(os.environ["DAISY_TEST_PIECES"]) if "DAISY_TEST_PIECES" in os.environ else None,
                    inter_matrix=sp.coo_matrix((np.ones(NNZ, np.float32), (gu, gi)), shape=(U, I)))
    model = LightGCN(cfg)
    model.fit(get_dataloader(BasicDataset(samples), batch_size=512, shuffle=True, num_workers=0))
    ue, ie = model.forward()
    return (np.array(model.epoch_losses), model.embed_user.weight.detach().cpu().numpy(),
            model.embed_item.weight.detach().cpu().numpy(), ue.cpu().numpy(), ie.cpu().numpy())


def _worker(rank, world, port, out_dir, backend):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if backend == "nccl":
        torch.cuda.set_device(rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    losses, P, Q, ue, ie = _train("auto")
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), losses=losses, P=P, Q=Q, ue=ue, ie=ie)
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _check(tmp_path, world):
    losses, P, Q, ue, ie = _train(False)                      # single process, local products
    outs = [np.load(os.path.join(str(tmp_path), f"r{r}.npz")) for r in range(world)]
    for o in outs:
        # the row blocks are bit-identical to the local product ('sorted': row-owner sums in stored order); the
        # element-wise passes around them round differently in the last place, Adam carries that along
        np.testing.assert_allclose(o["losses"], losses, rtol=1e-7)
        np.testing.assert_allclose(o["P"], P, atol=2e-5)
        np.testing.assert_allclose(o["Q"], Q, atol=2e-5)
        np.testing.assert_allclose(o["ue"], ue, atol=2e-5)
        np.testing.assert_array_equal(o["P"], outs[0]["P"])          # the replicas stay bit-identical
        np.testing.assert_array_equal(o["Q"], outs[0]["Q"])
    assert losses[1] < losses[0]


@pytest.mark.parametrize("pieces", [1, 3])
def test_row_sharded_training_on_three_ranks(tmp_path, pieces, monkeypatch):
    """pieces > 1: every layer's row block in sub-blocks, sub-block k all-gathered on a side stream while sub-block k+1
    is reduced; the node count (1150) is divisible neither by 3 ranks nor by 3 x 3 sub-blocks (padded tails)"""
    monkeypatch.setenv("DAISY_TEST_PIECES", str(pieces))
    world = 3
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), "gloo"), nprocs=world, join=True)
    _check(tmp_path, world)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs one GPU per rank (this box has one)")
def test_row_sharded_training_on_rccl(tmp_path):
    world = min(torch.cuda.device_count(), 8)
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), "nccl"), nprocs=world, join=True)
    _check(tmp_path, world)
