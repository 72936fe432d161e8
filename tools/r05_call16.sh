cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/r05/tests16.txt 2>&1; tail -5 gpurun_out/r05/tests16.txt
python tools/bench_neumf.py --no-cpu > gpurun_out/r05/bench_neumf.txt 2>&1; grep -E "\"B\"" gpurun_out/r05/bench_neumf.txt | cut -c1-170
