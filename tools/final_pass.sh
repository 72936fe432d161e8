cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/r05/tests_final.txt 2>&1; tail -3 gpurun_out/r05/tests_final.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/measure_round5.sh > gpurun_out/measure5c.txt 2>&1; tail -8 gpurun_out/measure5c.txt
python tools/bench_neumf.py --no-cpu > gpurun_out/r05/bench_neumf.txt 2>&1; grep -E "\"B\"" gpurun_out/r05/bench_neumf.txt | cut -c1-150
