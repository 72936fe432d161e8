#!/usr/bin/env python
"""The one-pass partition (csrc/bpr_staged.hip::k_part_onepass; DAISY_PLAN_ONEPASS=1: positions parked in LDS, records
loaded per sub-tile; =2: records front-loaded into registers; =3: three launches with nothing parked, the scatter walks
the positions itself) against the default three-launch plan build (count with parked positions / scan / scatter), which
the GPU tests pin to the oracle: every batch of every build must be identical, record for record.
Also times them.   timeout 60 python tools/r03_onepass_check.py [quick] [flags, e.g. 0,1,3]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from daisyrec_amd import ops  # noqa: E402

dev = torch.device("cuda")
quick = "quick" in sys.argv[1:]
FLAGS = next((a.split(",") for a in sys.argv[1:] if a[0].isdigit()), ["0", "1", "2", "3"])
assert FLAGS[0] == "0", "the three-launch build is the reference of the comparison"
# (n, U, I, B, order, pointwise): a partial last tile, a last batch that is not full, > 64 tiles (several look-back
# windows), one batch only, the identity order (every tile feeds few buckets)
CASES = [(300_001, 5000, 3000, 16384, "feistel", False),
         (1_500_000, 40_000, 9000, 65536, "feistel", False),
         (70_000, 900, 1100, 70_000, "feistel", False),
         (200_000, 3000, 2000, 4096, "identity", False),
         (400_000, 3000, 2000, 32768, "feistel", True)]
if not quick:
    CASES.append((20_000_000, 1_000_000, 100_000, 1 << 21, "feistel", False))


def batches(plan, nb, B, pointwise):
    out = []
    for k in range(nb):
        u, i, j, ei, es, _ = plan.read_batch(k, B)
        ne = u.shape[0] if pointwise else 2 * u.shape[0]          # point-wise rows carry one entry each
        out.append([u.clone(), i.clone(), j.clone(), ei[:ne].clone(), es[:ne].clone()])
    return out


ok = True
for n, U, I, B, order, pointwise in CASES:
    g = torch.Generator(device=dev)
    g.manual_seed(n)
    u = torch.randint(0, U, (n,), device=dev, generator=g, dtype=torch.int32).sort().values
    i = torch.randint(0, I, (n,), device=dev, generator=g, dtype=torch.int32)
    j = torch.randint(0, 2 if pointwise else I, (n,), device=dev, generator=g, dtype=torch.int32)
    triples = torch.stack([u, i, j], 1).contiguous()
    index = ops.TrainIndex(triples, U, I, user_sorted=True, pointwise=pointwise)
    res, ms = {}, {}
    for flag in FLAGS:
        os.environ["DAISY_PLAN_ONEPASS"] = flag
        plan = ops.EpochPlan(n, U, I, device=dev)
        plan.build_indexed(index, B, order=order, seed=3, epoch=1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for e in range(3):
            plan.build_indexed(index, B, order=order, seed=3, epoch=2 + e)
        torch.cuda.synchronize()
        ms[flag] = (time.perf_counter() - t0) / 3 * 1e3
        plan.build_indexed(index, B, order=order, seed=3, epoch=1)
        res[flag] = batches(plan, plan.num_batches, B, pointwise)
        plan.close()
    same = {f: all(torch.equal(a, b) for ka, kb in zip(res["0"], res[f]) for a, b in zip(ka, kb)) for f in FLAGS[1:]}
    ok &= all(same.values())
    print(f"n={n} B={B} {order}{' pointwise' if pointwise else ''}: three-launch {ms['0']:.3f} ms  "
          + "  ".join(f"variant[{f}] {ms[f]:.3f} ms identical={same[f]}" for f in FLAGS[1:]), flush=True)
    index.close()
os.environ["DAISY_PLAN_ONEPASS"] = "0"
print("ONEPASS_OK" if ok else "ONEPASS_MISMATCH")
sys.exit(0 if ok else 1)
