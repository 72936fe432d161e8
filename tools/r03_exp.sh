bash $GRAFT_REPO_ROOT/tools/measure_round3.sh
