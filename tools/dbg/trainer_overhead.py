"""Cost of the multi-GPU step sequence (UserShardedBprTrainer) vs the single-call step, on one GPU
with a world-size-1 RCCL group: what the N>1 bench path pays besides the collectives themselves."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.distributed as dist
import bench
from daisyrec_amd import ops
from daisyrec_amd.sharding import UserShardedBprTrainer
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
U, I, n, d, B = 1_000_000, 100_000, 50_000_000, 64, 1 << 21
tri = bench.synth_triples(U, I, n, 2022, dev)
n = tri.shape[0]
torch.manual_seed(0)
P = torch.empty(U, d, device=dev).normal_(0, 0.01); Q = torch.empty(I, d, device=dev).normal_(0, 0.01)
ctx = ops.BprContext(B, d, U, I, device=dev)
plan = ops.EpochPlan(n, U, I, device=dev).build(tri, B, order="feistel", seed=1, epoch=0, user_sorted=True)
tr = UserShardedBprTrainer(ctx, P, Q, 0, 0.01, 1e-3, 1e-3, item_mode=ops.ITEM_MODES["chunked"])
nb = n // B
def run(fn):
    for k in range(3): fn(k)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(nb): fn(k)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / nb, (time.perf_counter() - t0) / nb * 1e3
def single(k):
    ctx.set_batch_from_plan(plan, k); ctx.sgd_step(P, Q, 0.01, 1e-3, 1e-3, item_mode=2)
print("single-call step   gpu ms %.3f  wall ms %.3f" % run(single))
print("trainer (world 1)  gpu ms %.3f  wall ms %.3f" % run(lambda k: tr.step_from_plan(plan, k)))
dist.destroy_process_group()
