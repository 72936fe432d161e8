# round-3 measurement pass (run on the GPU box through gpurun): bench (primary + secondary + CPU baseline), kernel stats,
# PMC traffic, the p_stream A/B at configs[2] table shapes, Adam
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03
mkdir -p $O
timeout 400 python $R/bench.py > $O/bench.json 2> $O/bench.err
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o mf -- python $R/bench.py --no-cpu-baseline --no-secondary > $O/prof_bench.log 2>&1
python $R/tools/rocprof_summary.py $O/prof > $O/mf_kernel_summary.txt 2>/dev/null
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/mf_kernel_stats.csv 2>/dev/null
rm -rf $O/prof
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof2 -o mf -- python $R/bench.py --no-cpu-baseline --workload c3 --nnz 100000000 > $O/prof_bench_c3s.log 2>&1
python $R/tools/rocprof_summary.py $O/prof2 > $O/mf_kernel_summary_c3s.txt 2>/dev/null
rm -rf $O/prof2
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_rd -o rd -- python $R/bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-secondary > /dev/null 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_wr -o wr -- python $R/bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-secondary > /dev/null 2>&1
python $R/tools/pmc_traffic.py $O/pmc_rd $O/pmc_wr $O/pmc_traffic.json 2097152 "profiles/r03_pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py --steps 12 (round 3 kernels: staged step with nontemporal stage stores + Q window, partitioned plan)" > /dev/null
rm -rf $O/pmc_rd $O/pmc_wr
cd $R
if [ -n "$PSTREAM_AB" ]; then for ps in 0 1 0 1; do TAG=pstream$ps DAISY_STAGED_PSTREAM=$ps timeout 100 python tools/r03_probe.py c3s 40 2>&1 | grep "^\[" ; done | tee $O/pstream_ab.txt; fi
head -14 $O/mf_kernel_summary.txt | cut -c1-64,100-170
python - <<PY
import json
d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1])
print('primary', round(d['value']/1e9,3),'G/s', round(d['ms_per_step'],4),'ms/step frac',round(d['roofline']['frac'],3))
s=d.get('secondary'); 
if s: print('secondary', round(s['value']/1e9,3),'G/s', round(s['ms_per_step'],4), 'frac', round(s['roofline']['frac'],3))
c=d.get('cpu_baseline');
if c: print('cpu', c['value'], c['reference_default_batch'])
t=json.load(open('$O/pmc_traffic.json')); print('traffic/step GB', t['hbm_bytes_per_step']/1e9, 'B/inter', t['hbm_bytes_per_interaction'])
PY
