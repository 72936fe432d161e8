cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_staged.py tests/test_gpu_hardening.py -q -m gpu -s -k "c2_scale or merge or launch_forms or staged_epoch" > $O/tests8.txt 2>&1
grep -E "c2-scale|passed|failed" $O/tests8.txt
bash tools/pmc_plan.sh > $O/pmc_plan.txt 2>&1; cat $O/pmc_plan.txt
python tools/sweep_batch.py 16384 32768 65536 > $O/sweep8.txt 2>&1; cat $O/sweep8.txt
