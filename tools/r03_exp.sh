cd $GRAFT_REPO_ROOT
O=gpurun_out/r03
timeout 150 python tools/sweep_batch.py 2>/dev/null | tee $O/batch_sweep.txt
timeout 200 python tools/sweep_factors.py 2>/dev/null | tee $O/factor_sweep.txt
