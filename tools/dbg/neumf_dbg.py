import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, torch
from daisyrec_amd import ops
from oracle import neumf_numpy as NO
g = np.load("tests/golden/kat_neumf.npz")
U, I, d, L = (int(x) for x in g["mlsgd/meta"])
names = NO.param_names(L)
p_np = {k: g[f"mlsgd/{k}0"] for k in names}
for B in (256, 27, 64):
    b = g["ml/samples"][:B]
    want_loss, want = NO.neumf_grad(p_np, b[:, 0], b[:, 1], b[:, 2], 1e-3, 1e-3, L)
    p = {k: torch.as_tensor(v).cuda() for k, v in p_np.items()}
    grads = {k: torch.zeros_like(v) for k, v in p.items()}
    ctx = ops.NeumfContext(512, d, L, U, I)
    ctx.step_grads(p, grads, *(torch.as_tensor(b[:, k].copy()).cuda() for k in range(3)), 0, 1e-3, 1e-3)
    print(B, "loss", float(ctx.stats[11].cpu()), want_loss, "stats", ctx.stats.cpu().numpy()[:17].round(4))
    for k in names:
        e = np.abs(grads[k].cpu().numpy() - want[k]).max()
        print("  ", k, "max grad err", e, "ref max", np.abs(want[k]).max())
    ctx.close()
