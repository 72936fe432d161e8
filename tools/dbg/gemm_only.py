import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from daisyrec_amd import ops
M, N, K = 131072, 256, 512
A = torch.randn(M, K, device="cuda"); B = torch.randn(N, K, device="cuda")
for _ in range(3):
    ops.gemm_nt(A, B)
torch.cuda.synchronize()
