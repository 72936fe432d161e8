cd $GRAFT_REPO_ROOT
timeout 280 python -m pytest tests/test_gpu_staged.py -q -m gpu -k "fm_epoch" 2>&1 | grep -v "^$" | tail -6
