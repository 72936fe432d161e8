"""`torch.ops.daisyrec.*`: the C ABI registered as PyTorch custom ops on the HIP dispatch key
(csrc/torch_ops.cpp, a shim without logic; BASELINE.json north_star / SURVEY.md section 8b).

    import daisyrec_amd.torch_ops                      # loads lib/libdaisyrec_torch_ops.so
    loss = torch.ops.daisyrec.bpr_mf_step(P, Q, u, i, j, 0.01, 1e-3, 1e-3, 1e-10, 0)
    ids  = torch.ops.daisyrec.mf_rank_topk(P, Q, us, cands, 50)

The ctypes binding (`daisyrec_amd.ops`) and these ops call the same entry points; the model mirrors use the
ctypes route (it also reaches the plan / phase / multi-GPU entry points that are not tensor-in tensor-out)."""
import os

import torch

from . import _native  # noqa: F401  (maps libdaisyrec_hip.so first; fails loudly when it is missing)

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libdaisyrec_torch_ops.so")
if not os.path.exists(LIB_PATH):
    raise ImportError(f"{LIB_PATH} is missing: build it with `make -C daisyrec_amd/csrc` "
                      "(or `python -c 'import __graft_entry__ as g; g.build()'`)")
torch.ops.load_library(LIB_PATH)
OPS = ("mf_predict", "mf_rank_topk", "mf_full_rank", "sample_uniform_neg", "bpr_mf_step")
