cd $GRAFT_REPO_ROOT
timeout 280 python -m pytest tests/test_gpu_lightgcn_dist.py tests/test_gpu_lightgcn.py -q -m gpu 2>&1 | grep -v "^$" | tail -12
