#!/usr/bin/env python
"""bf16-storage GEMM alone (for rocprofv3 PMC passes): python tools/gemm_only.py [M N K reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from daisyrec_amd import ops

M, N, K, reps = (int(x) for x in (sys.argv[1:5] + ["524288", "256", "512", "6"][len(sys.argv) - 1:]))
A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
B = torch.randn(N, K, device="cuda").to(torch.bfloat16)
for _ in range(reps):
    ops.gemm_nt_bf16(A, B)
torch.cuda.synchronize()
