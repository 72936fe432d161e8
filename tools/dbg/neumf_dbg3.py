import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
from daisyrec_amd import ops
import daisyrec_amd.ops as O2
from test_gpu_neumf import _neumf_cfg
from daisyrec_amd.model.NeuMFRecommender import NeuMF
from daisyrec_amd.utils.dataset import BasicDataset, get_dataloader
g = np.load("tests/golden/kat_neumf.npz")
cfg, L = _neumf_cfg(g, "mlsgd")
print({k: cfg[k] for k in ("lr","reg_1","reg_2","optimizer","dropout","factors","num_layers","batch_size","epochs")})
torch.manual_seed(int(g["mlsgd/seed"]))
model = NeuMF(cfg)
loader = get_dataloader(BasicDataset(g["ml/samples"]), batch_size=256, shuffle=True, num_workers=4)
torch.set_rng_state(torch.from_numpy(g["mlsgd/rng_state_before_fit"]))
orig = O2.NeumfContext.step_grads
cnt = [0]
def spy(self, p, grads, u, i, j, *a, **k):
    orig(self, p, grads, u, i, j, *a, **k)
    if cnt[0] < 4 or cnt[0] in (100, 200, 306):
        print("step", cnt[0], float(self.stats[11].cpu()), u[:4].tolist(), i[:4].tolist(), a, k)
    cnt[0] += 1
O2.NeumfContext.step_grads = spy
model.fit(loader)
print(model.epoch_losses, g["mlsgd/epoch_losses"], "steps", cnt[0])
