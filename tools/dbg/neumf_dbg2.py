import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
from daisyrec_amd import ops
from oracle import neumf_numpy as NO
g = np.load("tests/golden/kat_neumf.npz")
U, I, d, L = (int(x) for x in g["mlsgd/meta"])
names = NO.param_names(L)
samples = g["ml/samples"]; n = len(samples); B = 256
torch.set_rng_state(torch.from_numpy(g["mlsgd/rng_state_before_fit"]))
torch.empty((), dtype=torch.int64).random_()
gen = torch.Generator(); gen.manual_seed(int(torch.empty((), dtype=torch.int64).random_().item()))
perm = torch.randperm(n, generator=gen).numpy()
p_np = {k: g[f"mlsgd/{k}0"].copy() for k in names}
p = {k: torch.as_tensor(v).cuda() for k, v in p_np.items()}
flat = torch.cat([p[k].reshape(-1) for k in names]).contiguous()
off = 0
for k in names:
    nn_ = p[k].numel(); p[k] = flat[off:off+nn_].view(p[k].shape); off += nn_
gflat = torch.zeros_like(flat); grads = {}; off = 0
for k in names:
    nn_ = p[k].numel(); grads[k] = gflat[off:off+nn_].view(p[k].shape); off += nn_
ctx = ops.NeumfContext(512, d, L, U, I)
tri = torch.as_tensor(samples).cuda()
order = tri[torch.as_tensor(perm).cuda()]
tot_g = tot_o = 0.0
for st in range(6):
    s = st * B
    idx = perm[s:s+B]
    lo, gr = NO.neumf_grad(p_np, samples[idx,0], samples[idx,1], samples[idx,2], 1e-3, 1e-3, L)
    p_np = {k: (np.asarray(p_np[k], np.float64) - 0.01*gr[k]).astype(np.float32) for k in names}
    rows = order[s:s+B]
    u,i,j = (rows[:,k].contiguous() for k in range(3))
    ctx.step_grads(p, grads, u, i, j, 0, 1e-3, 1e-3)
    lg = float(ctx.stats[11].cpu())
    ops.sgd_dense(flat, gflat, 0.01)
    err = max(np.abs(p[k].cpu().numpy()-p_np[k]).max() for k in names)
    print(st, lg, lo, "param err", err)
