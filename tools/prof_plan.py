#!/usr/bin/env python
"""Plan builds only (for rocprofv3 --kernel-trace): 3 partitioned builds at C2 shapes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from daisyrec_amd import ops  # noqa: E402

U, I, nnz, B = 1_000_000, 100_000, 50_000_000, 1 << 21
tri = bench.synth_triples(U, I, nnz, 2022, torch.device("cuda"))
index = ops.TrainIndex(tri, U, I, user_sorted=True)
plan = ops.EpochPlan(tri.shape[0], U, I)
for e in range(3):
    plan.build_indexed(index, B, order="feistel", seed=1, epoch=e)
torch.cuda.synchronize()
