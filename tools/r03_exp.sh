cd $GRAFT_REPO_ROOT
timeout 200 python -m pytest tests/test_gpu_fit_dist.py -x -q -m gpu -k "pointwise or adam" 2>&1 | grep -v "^$" | tail -30
