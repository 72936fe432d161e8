cd $GRAFT_REPO_ROOT
timeout 560 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fm.py tests/test_gpu_staged.py tests/test_gpu_property.py tests/test_gpu_hardening.py tests/test_gpu_plan.py tests/test_gpu_torch_ops.py tests/test_gpu_fit_dist.py tests/test_gpu_dist.py -x -q -m gpu --durations=8 2>&1 | tail -30
