cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
python tools/time_bench_legs.py > gpurun_out/r05/bench_legs.txt 2>&1; cat gpurun_out/r05/bench_legs.txt
