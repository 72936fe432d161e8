"""Host logic of the mirror classes (no device): construction from the reference's config keys, parameter
initialisation bit-identical to the reference's (same module order => same torch RNG stream; the golden
files hold the reference's initial parameters), optimiser / loss name handling, and the refusal to run
without a HIP device (no CPU fallback)."""
import os

import numpy as np
import pytest
import scipy.sparse as sp
import torch

from conftest import mf_config

HERE = os.path.dirname(os.path.abspath(__file__))
G = lambda name: np.load(os.path.join(HERE, "golden", name))     # noqa: E731
NO_GPU = not torch.cuda.is_available()


def test_fm_init_matches_the_reference():
    from daisyrec_amd.model.FMRecommender import FM
    g = G("kat_fm.npz")
    torch.manual_seed(int(g["ml/seed"]))
    m = FM(mf_config(user_num=int(g["ml/user_num"]), item_num=int(g["ml/item_num"]), factors=int(g["ml/factors"]),
                     algo_name="fm"))
    np.testing.assert_array_equal(m.embed_user.weight.detach().numpy(), g["ml/P0"])
    np.testing.assert_array_equal(m.embed_item.weight.detach().numpy(), g["ml/Q0"])
    assert float(m.u_bias.weight.detach().abs().max()) == 0 and float(m.i_bias.weight.detach().abs().max()) == 0
    assert m.optimizer == "sgd" and m.initializer == "normal"


def test_neumf_init_matches_the_reference_and_model_names():
    from daisyrec_amd.model.NeuMFRecommender import NeuMF
    g = G("kat_neumf.npz")
    U, I, d, L = (int(x) for x in g["ml/meta"])
    cfg = mf_config(user_num=U, item_num=I, factors=d, num_layers=L, dropout=0.0, model_name="NeuMF", GMF_model=None,
                    MLP_model=None, algo_name="neumf")
    torch.manual_seed(int(g["ml/seed"]))
    m = NeuMF(cfg)
    for k, p in m._named().items():
        np.testing.assert_array_equal(p.detach().numpy(), g[f"ml/{k}0"], err_msg=k)
    assert m.optimizer == "adam" and m.initializer == "xavier_normal"
    assert NeuMF({**cfg, "model_name": "GMF"}).predict_layer.in_features == d
    assert NeuMF({**cfg, "model_name": "MLP"}).predict_layer.in_features == d
    pre = NeuMF({**cfg, "model_name": "NeuMF-pre", "GMF_model": NeuMF({**cfg, "model_name": "GMF"}),
                 "MLP_model": NeuMF({**cfg, "model_name": "MLP"})})
    assert pre.predict_layer.in_features == 2 * d


def test_lightgcn_and_item2vec_init_match_the_reference():
    from daisyrec_amd.model.Item2VecRecommender import Item2Vec
    from daisyrec_amd.model.LightGCNRecommender import LightGCN
    g = G("kat_lightgcn.npz")
    U, I, d, L = (int(x) for x in g["ml/meta"])
    gu, gi = g["ml/train_users"], g["ml/train_items"]
    torch.manual_seed(int(g["ml/seed"]))
    m = LightGCN(mf_config(user_num=U, item_num=I, factors=d, num_layers=L, algo_name="lightgcn", reg_1=0.0, reg_2=0.0,
                           inter_matrix=sp.coo_matrix((np.ones(len(gu), np.float32), (gu, gi)), shape=(U, I))))
    np.testing.assert_array_equal(m.embed_user.weight.detach().numpy(), g["ml/P0"])
    np.testing.assert_array_equal(m.embed_item.weight.detach().numpy(), g["ml/Q0"])
    assert m.optimizer == "adam" and m.initializer == "xavier_uniform" and m.item_mode == "chunked"
    g = G("kat_item2vec.npz")
    U, I, d = (int(x) for x in g["ml/meta"])
    torch.manual_seed(int(g["ml/seed"]))
    m = Item2Vec(mf_config(user_num=U, item_num=I, factors=d, train_ur={}, algo_name="item2vec"))
    np.testing.assert_array_equal(m.shared_embedding.weight.detach().numpy(), g["ml/S0"])
    np.testing.assert_array_equal(m.user_embedding.weight.detach().numpy(), g["ml/Uemb0"])
    assert m.loss_type == "CL" and m.optimizer == "adam"


@pytest.mark.skipif(not NO_GPU, reason="host-only behaviour")
def test_models_refuse_to_run_without_a_device():
    from daisyrec_amd.model.FMRecommender import FM
    from daisyrec_amd.model.Item2VecRecommender import Item2Vec
    from daisyrec_amd.model.NeuMFRecommender import NeuMF
    from daisyrec_amd.utils.dataset import BasicDataset, get_dataloader
    loader = get_dataloader(BasicDataset(np.zeros((8, 3), np.int32)), batch_size=4, shuffle=False, num_workers=0)
    models = [FM(mf_config(user_num=5, item_num=5, factors=8)),
              NeuMF(mf_config(user_num=5, item_num=5, factors=8, num_layers=2, dropout=0.0, model_name="NeuMF",
                              GMF_model=None, MLP_model=None)),
              Item2Vec(mf_config(user_num=5, item_num=5, factors=8, train_ur={}))]
    for m in models:
        with pytest.raises(RuntimeError, match="no HIP device"):
            m.fit(loader)
        with pytest.raises(RuntimeError, match="no HIP device"):
            m.full_rank(0)


def test_optimizer_and_loss_names_follow_the_reference():
    from daisyrec_amd.model.FMRecommender import FM
    m = FM(mf_config(user_num=5, item_num=5, factors=8, optimizer="nonsense"))
    assert m._resolve_optimizer() == "adam"                 # unknown -> Adam with a log line (AbstractRecommender.py:63-65)
    assert FM(mf_config(user_num=5, item_num=5, factors=8, optimizer="rmsprop"))._resolve_optimizer() == "rmsprop"
    with pytest.raises(RuntimeError, match="SparseAdam"):
        FM(mf_config(user_num=5, item_num=5, factors=8, optimizer="sparse_adam"))._resolve_optimizer()
    with pytest.raises(NotImplementedError):
        m._build_criterion("XX")


def test_native_knobs_are_parsed_strictly():
    """config['lazy_adam'] from yaml / CLI strings (bool('false') is True), unknown NeuMF precisions"""
    from daisyrec_amd.model.AbstractRecommender import _parse_lazy_adam
    from daisyrec_amd.model.MFRecommender import MF
    from daisyrec_amd.model.NeuMFRecommender import NeuMF
    for v, want in ((True, True), (False, False), ("true", True), ("False", False), ("0", False), ("1", True),
                    (0, False), (1, True), ("auto", "auto"), ("AUTO", "auto"), ("off", False), ("yes", True)):
        assert _parse_lazy_adam(v) == want, v
    for bad in ("maybe", "", None, 2, "lazy"):
        with pytest.raises(ValueError):
            _parse_lazy_adam(bad)
    assert MF(mf_config(user_num=30, item_num=20, lazy_adam="false")).lazy_adam is False
    with pytest.raises(ValueError):
        MF(mf_config(user_num=30, item_num=20, lazy_adam="sometimes"))
    base = dict(user_num=30, item_num=20, factors=8, num_layers=2, dropout=0.0, model_name="NeuMF", GMF_model=None,
                MLP_model=None, algo_name="neumf")
    assert NeuMF(mf_config(**base, precision="bf16")).precision == "bf16"
    with pytest.raises(ValueError, match="precision"):
        NeuMF(mf_config(**base, precision="fp16"))


def test_sharded_adam_table_matches_the_dense_optimisers_constants():
    """ops.ShardedAdam / ops.LazyAdam take their per-step constants from daisy_adam_lazy_table: lr / (1 - beta1^t) and
    sqrt(1 - beta2^t), the host arithmetic of daisy_adam_dense (torch.optim.Adam's bias corrections)"""
    import ctypes as C
    from daisyrec_amd import _native as N
    n = 12
    host = (C.c_float * (2 * (n + 1)))()
    N.check(N.lib.daisy_adam_lazy_table(0.01, 0.9, 0.999, n, host))
    t = np.frombuffer(host, dtype=np.float32).reshape(n + 1, 2)
    lr, b1, b2 = (float(np.float32(x)) for x in (0.01, 0.9, 0.999))        # the C entry takes floats
    for s in range(1, n + 1):
        assert t[s, 0] == np.float32(lr / (1.0 - b1 ** s)) and t[s, 1] == np.float32(np.sqrt(1.0 - b2 ** s)), s


def test_state_dict_of_a_table_on_a_row_pitch_is_contiguous(tmp_path):
    """a table homed on a row pitch is an [n, d] VIEW of a padded [n, d'] buffer; torch.save(state_dict())
    would write the whole padded storage (a third more bytes at d = 50, and not comparable with the reference's
    checkpoint).  The state dict carries contiguous [n, d] copies; loading one back gives the same weights."""
    import torch
    from daisyrec_amd.model.MFRecommender import MF
    cfg = {"gpu": "0", "logger": None, "lr": 0.01, "reg_1": 0.0, "reg_2": 0.0, "epochs": 1, "topk": 5, "user_num": 40,
           "item_num": 30, "factors": 50, "loss_type": "BPR", "optimizer": "sgd", "init_method": "default",
           "early_stop": False}
    m = MF(cfg)
    want = {k: v.clone() for k, v in m.state_dict().items()}
    for name in ("embed_user", "embed_item"):                   # what `_tables()` does on the device: re-home on 64 columns
        w = getattr(m, name).weight
        buf = torch.zeros(w.shape[0], 64)
        buf[:, :50].copy_(w.data)
        w.data = buf[:, :50]
        assert not w.data.is_contiguous()
    sd = m.state_dict()
    for k, v in sd.items():
        assert v.is_contiguous() and tuple(v.shape) == tuple(want[k].shape) and torch.equal(v, want[k])
    path = tmp_path / "mf.pt"
    torch.save(sd, path)
    assert path.stat().st_size < 1.15 * (40 + 30) * 50 * 4 + 4096          # the bare tables, not the padded storage
    m2 = MF(cfg)
    m2.load_state_dict(torch.load(path))
    assert torch.equal(m2.embed_user.weight.data, want["embed_user.weight"])
