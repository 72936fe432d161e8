cd $GRAFT_REPO_ROOT
timeout 560 python -m pytest tests -q -m gpu --durations=6 2>&1 | tail -22
