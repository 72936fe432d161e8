import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from daisyrec_amd import ops
dev = "cuda"
U, I, n, B = 1_000_000, 100_000, 50_000_000, int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
g = torch.Generator(device=dev); g.manual_seed(0)
u = torch.randint(0, U, (n,), device=dev, generator=g, dtype=torch.int32).sort().values
tri = torch.stack([u, torch.randint(0, I, (n,), device=dev, generator=g, dtype=torch.int32),
                   torch.randint(0, I, (n,), device=dev, generator=g, dtype=torch.int32)], 1).contiguous()
plan = ops.EpochPlan(n, U, I)
torch.cuda.synchronize()
print("MARK")
for e in range(4):
    plan.build(tri, B, order="feistel", seed=1, epoch=e, user_sorted=True)
torch.cuda.synchronize()
s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for e in range(4):
    plan.build(tri, B, order="feistel", seed=1, epoch=e, user_sorted=True)
t.record(); torch.cuda.synchronize()
print("plan ms", s.elapsed_time(t) / 4)
