# round-3 diagnostic pass 1: where does the step spend its time at configs[2] shapes on one GPU?
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03
mkdir -p $O
python $R/bench.py --no-cpu-baseline > $O/d1_bench_c2.json 2> $O/d1_bench_c2.err
python $R/bench.py --no-cpu-baseline --workload c3 --nnz 100000000 > $O/d1_bench_c3s.json 2> $O/d1_bench_c3s.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/d1_prof_c3s -o mf -- python $R/bench.py --no-cpu-baseline --workload c3 --nnz 100000000 --steps 40 > $O/d1_prof_c3s.log 2>&1
python $R/tools/rocprof_summary.py $O/d1_prof_c3s > $O/d1_c3s_kernel_summary.txt 2>/dev/null
rm -rf $O/d1_prof_c3s
cd $R
head -16 $O/d1_c3s_kernel_summary.txt | cut -c1-70,100-170
for f in $O/d1_bench_*.json; do echo "$f: $(python -c "
import json,sys
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); print(round(d['value']/1e9,3),'G/s', round(d['ms_per_step'],4),'ms/step frac',round(d['roofline']['frac'],3))
except Exception as e: print('ERR',e)
")"; done
