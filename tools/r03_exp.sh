cd $GRAFT_REPO_ROOT
timeout 400 python -m pytest tests/test_gpu_staged.py tests/test_gpu_property.py tests/test_gpu_hardening.py tests/test_gpu_parity.py tests/test_gpu_fm.py tests/test_gpu_plan.py tests/test_gpu_torch_ops.py -q -m gpu 2>&1 | grep -v "^$" | tail -8
