#!/usr/bin/env python
"""NeuMF steps at one precision level for rocprofv3 --kernel-trace: python tools/dbg/neumf_prof.py <level> [B]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.argv = [sys.argv[0]] + sys.argv[1:]
import torch  # noqa: E402

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench_neumf as bn  # noqa: E402

level = int(sys.argv[1]) if len(sys.argv) > 1 else 2
B = int(sys.argv[2]) if len(sys.argv) > 2 else 262144
bn.run(B, 6, bf16=level)
