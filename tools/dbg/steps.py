"""N chunked/fused SGD steps on one random C2-shaped batch (for rocprofv3 PMC passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from daisyrec_amd import ops
dev = "cuda"
U, I, B, d = 1_000_000, 100_000, 1 << 20, 64
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 2
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
g = torch.Generator(device=dev); g.manual_seed(0)
P = torch.randn(U, d, device=dev, generator=g) * 0.01
Q = torch.randn(I, d, device=dev, generator=g) * 0.01
tri = torch.stack([torch.randint(0, U, (B,), device=dev, generator=g, dtype=torch.int32),
                   torch.randint(0, I, (B,), device=dev, generator=g, dtype=torch.int32),
                   torch.randint(0, I, (B,), device=dev, generator=g, dtype=torch.int32)], 1).contiguous()
ctx = ops.BprContext(B, d, U, I)
ctx.set_batch_from_triples(tri)
for _ in range(steps):
    ctx.sgd_step(P, Q, 1e-9, 1e-3, 1e-3, item_mode=mode)
torch.cuda.synchronize()
print("done")
