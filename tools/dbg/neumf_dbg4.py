import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
from daisyrec_amd import ops
from oracle import neumf_numpy as NO
g = np.load("tests/golden/kat_neumf.npz")
U, I, d, L = (int(x) for x in g["mlsgd/meta"])
names = NO.param_names(L)
samples = torch.as_tensor(g["ml/samples"]).cuda(); n = len(samples); B = 256
p = {k: torch.as_tensor(g[f"mlsgd/{k}1"]).cuda() for k in names}     # trained parameters
ctx = ops.NeumfContext(512, d, L, U, I)
gen = torch.Generator(device="cuda"); gen.manual_seed(0)
bad = 0
for trial in range(300):
    idx = torch.randint(0, n, (B,), device="cuda", generator=gen)
    rows = samples[idx]
    u, i, j = (rows[:, k].contiguous() for k in range(3))
    outs = []
    for rep in range(2):
        grads = {k: torch.zeros_like(v) for k, v in p.items()}
        ctx.step_grads(p, grads, u, i, j, 0, 1e-3, 1e-3)
        outs.append((float(ctx.stats[11].cpu()), {k: v.clone() for k, v in grads.items()}))
    dl = abs(outs[0][0] - outs[1][0])
    dg = {k: float((outs[0][1][k] - outs[1][1][k]).abs().max()) for k in names}
    worst = max(dg.values())
    if dl > 1e-6 * abs(outs[0][0]) or worst > 1e-4:
        bad += 1
        print("trial", trial, "loss", outs[0][0], outs[1][0], {k: round(v, 6) for k, v in dg.items() if v > 1e-5})
print("bad trials", bad)
