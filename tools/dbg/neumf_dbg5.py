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
perm = torch.randperm(n, generator=gen)
tri = torch.as_tensor(samples).cuda()
order = tri[perm.cuda()]
ctx = ops.NeumfContext(512, d, L, U, I)
hist = []
for rep in range(6):
    shapes = {k: g[f"mlsgd/{k}0"].shape for k in names}
    flat = torch.cat([torch.as_tensor(g[f"mlsgd/{k}0"]).reshape(-1) for k in names]).cuda().contiguous()
    gflat = torch.zeros_like(flat)
    p, grads, off = {}, {}, 0
    for k in names:
        nn_ = int(np.prod(shapes[k])); p[k] = flat[off:off+nn_].view(shapes[k]); grads[k] = gflat[off:off+nn_].view(shapes[k]); off += nn_
    losses = torch.zeros(307, dtype=torch.float64, device="cuda")
    for st in range(307):
        rows = order[st*B:(st+1)*B]
        u,i,j = (rows[:,k].contiguous() for k in range(3))
        ctx.step_grads(p, grads, u, i, j, 0, 1e-3, 1e-3)
        losses[st] = ctx.stats[11]
        ops.sgd_dense(flat, gflat, 0.001)
    l = losses.cpu().numpy()
    hist.append(l)
    print(rep, "total", l.sum(), "l[100,150,200,306]", l[[100,150,200,306]])
h = np.stack(hist)
dev = np.abs(h - h[0]).max(0) / np.abs(h[0])
first = np.argmax(dev > 1e-4)
print("first step with >1e-4 rel deviation:", first, dev[max(first-3,0):first+5])
