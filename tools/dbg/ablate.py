import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from daisyrec_amd import ops
dev = "cuda"
U, I, B, d = 1_000_000, 100_000, 1 << 20, 64
g = torch.Generator(device=dev); g.manual_seed(0)
P = torch.randn(U, d, device=dev, generator=g) * 0.01
Q = torch.randn(I, d, device=dev, generator=g) * 0.01
tri = torch.stack([torch.randint(0, U, (B,), device=dev, generator=g, dtype=torch.int32),
                   torch.randint(0, I, (B,), device=dev, generator=g, dtype=torch.int32),
                   torch.randint(0, I, (B,), device=dev, generator=g, dtype=torch.int32)], 1).contiguous()
ctx = ops.BprContext(B, d, U, I)
ctx.set_batch_from_triples(tri)
def t(fn, it=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it
print("fused step ms", t(lambda: ctx.sgd_step(P, Q, 1e-9, 1e-3, 1e-3, item_mode=3)), "ablate", os.environ.get("DAISY_FUSED_ABLATE"))
