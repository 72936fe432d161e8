cd $GRAFT_REPO_ROOT
timeout 280 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fm.py -x -q -m gpu -k "kat or pointwise or fm_step or adam" 2>&1 | tail -25
