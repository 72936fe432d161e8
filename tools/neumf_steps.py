#!/usr/bin/env python
"""A few NeuMF steps at one precision level (for rocprofv3 --kernel-trace: bash tools/kstats.sh 24 python
tools/neumf_steps.py <level 0|1|2> [B])"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
level = int(sys.argv[1]) if len(sys.argv) > 1 else 2
B = int(sys.argv[2]) if len(sys.argv) > 2 else 262144
sys.argv = sys.argv[:1]
import bench_neumf as bn  # noqa: E402

bn.run(B, 6, bf16=level)
