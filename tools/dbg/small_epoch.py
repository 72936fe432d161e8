"""Launch-bound regime: ml-100k-sized MF epochs at the reference's default batch (256), with and without
the epoch hipGraph (DAISY_EPOCH_GRAPH=0/1)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from daisyrec_amd import ops
U, I, n, d, B = 943, 1152, 78363, 32, int(sys.argv[1]) if len(sys.argv) > 1 else 256
rng = np.random.default_rng(0)
tri = torch.as_tensor(np.stack([rng.integers(0, U, n), rng.integers(0, I, n), rng.integers(0, I, n)], 1).astype(np.int32)).cuda()
P = torch.randn(U, d, device="cuda") * 0.01; Q = torch.randn(I, d, device="cuda") * 0.01
ctx = ops.BprContext(B, d, U, I); plan = ops.EpochPlan(n, U, I)
for mode in ("sorted", "chunked"):
    losses = []
    for ep in range(6):
        plan.build(tri, B, order="feistel", seed=1, epoch=ep)
        ctx.epoch_acc.zero_()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ctx.fit_epoch_sgd(plan, P, Q, 0.01, 1e-3, 1e-3, item_mode=ops.ITEM_MODES[mode])
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        losses.append(float(ctx.epoch_acc[0].cpu()))
    print(mode, "B", B, "steps", plan.num_batches, "last epoch ms %.3f  us/step %.2f" % (dt * 1e3, dt * 1e6 / plan.num_batches), "loss", losses[0], losses[-1])
