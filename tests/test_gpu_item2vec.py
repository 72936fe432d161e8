"""Item2Vec widening (rest of SURVEY.md §8f rank 4) on the GPU against the golden vectors of the REAL
reference (tests/golden/kat_item2vec.npz) and the CPU oracle."""
import os

import numpy as np
import pytest
import torch

from conftest import mf_config
from oracle import item2vec_numpy as IV
from test_oracle_neumf import assert_params_close

pytestmark = pytest.mark.gpu
DEV = "cuda"
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def kat_i2v():
    return np.load(os.path.join(HERE, "golden", "kat_item2vec.npz"))


def _t(x):
    return torch.as_tensor(np.ascontiguousarray(x)).to(DEV)


def test_item2vec_kat_steps(kat_i2v):
    from daisyrec_amd import ops
    from daisyrec_amd.model.Item2VecRecommender import Item2Vec
    g = kat_i2v
    for name in g["names"]:
        name = str(name)
        U, I, d, B, ns = (int(x) for x in g[f"{name}/meta"])
        lr = float(g[f"{name}/lr"])
        model = Item2Vec(mf_config(user_num=U, item_num=I, factors=d, lr=lr, optimizer=str(g[f"{name}/optimizer"]),
                                   train_ur={}, epochs=1))
        S = _t(g[f"{name}/S0"])
        gA, gB = torch.zeros_like(S), torch.zeros_like(S)
        m, v = torch.zeros_like(S), torch.zeros_like(S)
        ctx = ops.BprContext(B, d, I, I)
        ctx.set_pointwise(True)
        is_adam = model.optimizer == "adam"
        for s in range(ns):
            model._step(ctx, S, gA, gB, _t(g[f"{name}/t"][s]), _t(g[f"{name}/c"][s]), _t(g[f"{name}/y"][s]))
            loss, ref = float(ctx.stats[7].cpu()), float(g[f"{name}/loss"][s])
            assert abs(loss - ref) <= 1e-5 * abs(ref), (name, s, loss, ref)
            if s == 0:                                  # the gradient itself against the oracle
                _, want = IV.item2vec_grad(g[f"{name}/S0"], g[f"{name}/t"][0], g[f"{name}/c"][0], g[f"{name}/y"][0])
                np.testing.assert_allclose(gA.cpu().numpy(), want, rtol=1e-4, atol=1e-6)
                assert float(gB.abs().max().cpu()) == 0.0
            if is_adam:
                ops.adam_dense(S, gA, m, v, lr, s + 1)
            else:
                ops.sgd_dense(S, gA, lr)
        assert_params_close({"S": S.cpu().numpy()}, {"S": g[f"{name}/S"]}, ("S",), name, 5e-6,
                            adam_lr=lr if is_adam else None, steps=ns, frac=0.98)
        ctx.close()


def test_item2vec_ml100k_through_the_dropin(kat_i2v):
    """run_examples/test.py --algo_name item2vec on ml-100k (item2vec.yaml: d=100, Adam lr 0.001; 50 batches of
    the reference sampler's triples) through Item2Vec.fit / rank."""
    from daisyrec_amd.model.Item2VecRecommender import Item2Vec
    from daisyrec_amd.utils.dataset import BasicDataset, CandidatesDataset, get_dataloader
    g = kat_i2v
    U, I, d = (int(x) for x in g["ml/meta"])
    ur = {}
    for u, i in zip(g["ml/ur_users"], g["ml/ur_items"]):
        ur.setdefault(int(u), set()).add(int(i))
    torch.manual_seed(int(g["ml/seed"]))
    model = Item2Vec(mf_config(user_num=U, item_num=I, factors=d, lr=float(g["ml/lr"]), epochs=1, train_ur=ur,
                               topk=int(g["ml/topk"]), algo_name="item2vec"))
    np.testing.assert_array_equal(model.shared_embedding.weight.detach().numpy(), g["ml/S0"])
    np.testing.assert_array_equal(model.user_embedding.weight.detach().numpy(), g["ml/Uemb0"])
    loader = get_dataloader(BasicDataset(g["ml/samples"]), batch_size=int(g["ml/batch_size"]), shuffle=True, num_workers=4)
    torch.set_rng_state(torch.from_numpy(g["ml/rng_state_before_fit"]))
    model.fit(loader)
    ref = float(g["ml/epoch_losses"][0])
    assert abs(model.epoch_losses[0] - ref) <= 1e-5 * abs(ref), (model.epoch_losses, ref)
    np.testing.assert_allclose(model.shared_embedding.weight.detach().cpu().numpy(), g["ml/S1"], atol=3e-5)
    np.testing.assert_allclose(model.user_embedding.weight.detach().cpu().numpy(), g["ml/Uemb1"], rtol=1e-4, atol=2e-4)
    ucands = [[int(u), c] for u, c in zip(g["ml/test_u"], g["ml/cands"])]
    preds = model.rank(get_dataloader(CandidatesDataset(ucands), batch_size=128, shuffle=False, num_workers=0))
    assert preds.dtype == np.float32 and (preds == g["ml/preds"]).all(axis=1).mean() > 0.9
    full = np.stack([model.full_rank(int(u)) for u in g["ml/test_u"][:8]])
    assert (full == g["ml/full8"]).mean() > 0.9
    pp = np.array([model.predict(int(u), 5) for u in g["ml/test_u"][:8]])
    np.testing.assert_allclose(pp, g["ml/predict8"], rtol=1e-3, atol=1e-5)
    b = g["ml/samples"][:256]
    loss = float(model.calc_loss([torch.from_numpy(b[:, k].copy()) for k in range(3)]).cpu())
    want, _ = IV.item2vec_grad(model.shared_embedding.weight.detach().cpu().numpy(), b[:, 0], b[:, 1], b[:, 2])
    assert abs(loss - want) <= 1e-5 * abs(want)


def test_skipgram_sampler_on_the_device():
    """SkipGramNegativeSampler (sampler.py:105-160) mirror: bit-exact against the oracle's restatement (same Philox
    draws); the (target, context, 1) rows are what the reference's window loop emits, in its order; negatives are
    never one of the user's train items; users with short sequences, a user without rows, window larger than a
    sequence."""
    import pandas as pd
    from daisyrec_amd.utils.sampler import SkipGramNegativeSampler
    from oracle import bpr_mf_numpy as O
    rng = np.random.default_rng(4)
    U, I, w = 30, 50, 3
    users = rng.integers(0, U, 400)
    users[users == 7] = 8                                    # user 7 has no rows
    users[:2] = 29                                           # a two-element sequence
    items = rng.integers(0, I, 400)
    df = pd.DataFrame({"user": users, "item": items, "rating": 1})
    ur = {u: set() for u in range(U)}
    for u, i in zip(users, items):
        ur[int(u)].add(int(i))
    cfg = {"UID_NAME": "user", "IID_NAME": "item", "item_num": I, "user_num": U, "train_ur": ur, "context_window": w,
           "rho": 1e-5, "seed": 2022}
    smp = SkipGramNegativeSampler(df, cfg)
    got = smp.sampling()
    seqs = [(int(u), [int(x) for x in g_]) for u, g_ in df.groupby("user")["item"].agg(list).items()]
    rows = {u: np.array(sorted(s), dtype=np.int32) for u, s in ur.items()}
    want = O.skipgram_samples(seqs, rows, I, w, 2022, SkipGramNegativeSampler.STREAM)
    assert got.dtype == np.int64 and got.shape == want.shape
    np.testing.assert_array_equal(got, want)
    # the reference's own loop for the positives (sampler.py:139-149), and the negative property
    ref_pos = []
    for u, seq in seqs:
        for i in range(len(seq)):
            j = i - w
            while j <= i + w and j < len(seq):
                if j >= 0 and j != i:
                    ref_pos.append([seq[i], seq[j], 1])
                j += 1
    np.testing.assert_array_equal(got[got[:, 2] == 1], np.array(ref_pos))
    assert (got[:, 2] == 0).sum() == (got[:, 2] == 1).sum()
    # every negative row follows its target's positives and avoids the user's items
    k = 0
    for u, seq in seqs:
        for i in range(len(seq)):
            c = min(i, w) + min(len(seq) - 1 - i, w)
            blk = got[k:k + 2 * c]
            assert (blk[:, 0] == seq[i]).all() and (blk[:c, 2] == 1).all() and (blk[c:, 2] == 0).all()
            assert not (set(blk[c:, 1].tolist()) & ur[u])
            k += 2 * c
    assert k == len(got)
