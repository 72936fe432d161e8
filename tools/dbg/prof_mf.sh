cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o mf -- python $R/bench.py --no-cpu-baseline > $R/gpurun_out/prof_bench.log 2>&1
python $R/tools/rocprof_summary.py $R/gpurun_out/prof > $R/gpurun_out/mf_kernel_summary.txt 2>/dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_rd -o rd -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_wr -o wr -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
python $R/tools/pmc_traffic.py $R/gpurun_out/pmc_rd $R/gpurun_out/pmc_wr $R/gpurun_out/pmc_traffic.json
cd $R && python bench.py 2>&1 | tail -1 > gpurun_out/bench_final.json
head -12 gpurun_out/mf_kernel_summary.txt | cut -c1-60,100-160
cat gpurun_out/pmc_traffic.json | head -c 600
