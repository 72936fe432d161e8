# round-6 measurement pass (run on the GPU box through gpurun): bench (primary + secondary + CPU baseline), kernel stats
# and PMC traffic at BOTH shapes (configs[1] and configs[2] table shapes), the full-size configs[2] set on one GPU
# (FULL_C3=1), sweeps.  Output under gpurun_out/r06m; copy what is worth keeping to profiles/r05_*.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06m
mkdir -p $O
timeout 600 python $R/bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
C3S="--workload c3 --nnz 100000000"
for tag in c2 c3s; do
  if [ $tag = c2 ]; then W="--no-secondary"; else W="$C3S"; fi
  timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$tag -o mf -- python $R/bench.py --no-cpu-baseline --no-extras $W > $O/prof_bench_$tag.log 2>&1
  python $R/tools/rocprof_summary.py $O/prof_$tag > $O/kernel_summary_$tag.txt 2>/dev/null
  cp $(find $O/prof_$tag -name "*kernel_stats.csv" | head -1) $O/kernel_stats_$tag.csv 2>/dev/null
  rm -rf $O/prof_$tag
  timeout 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_rd -o rd -- python $R/bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-extras $W > /dev/null 2>&1
  timeout 240 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_wr -o wr -- python $R/bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-extras $W > /dev/null 2>&1
  python $R/tools/pmc_traffic.py $O/pmc_rd $O/pmc_wr $O/pmc_traffic_$tag.json 2097152 "profiles/r06_pmc_traffic_$tag.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py --steps 12 $W (round 6: the MF step kernels are round 5's)" > /dev/null
  rm -rf $O/pmc_rd $O/pmc_wr
done
# NeuMF (configs[3] shapes): whole steps per precision, kernel stats of the bf16 step, the tower's counters and phase cycles
python $R/tools/bench_neumf.py > $O/bench_neumf.txt 2>&1
bash $R/tools/kstats.sh 60 python $R/tools/neumf_steps.py 2 262144 > $O/neumf_kernel_summary_bf16.txt 2>&1
bash $R/tools/kstats.sh 60 python $R/tools/neumf_steps.py 0 262144 > $O/neumf_kernel_summary_fp32.txt 2>&1
bash $R/tools/pmc_tower.sh > $O/pmc_tower.txt 2>&1
# the reference's own batch sizes
python $R/tools/adam_small.py > $O/small_batch.txt 2>&1
python $R/tools/neumf_small.py >> $O/small_batch.txt 2>&1
python $R/tools/neumf_small.py 24 2 256 0.0 >> $O/small_batch.txt 2>&1
bash $R/tools/kstats.sh 60 python $R/tools/neumf_small.py > $O/neumf_small_kernel_summary.txt 2>&1
cd $R
if [ -n "$FULL_C3" ]; then timeout 500 python bench.py --workload c3 --no-cpu-baseline > $O/bench_c3_full.json 2> $O/bench_c3_full.err; fi
head -14 $O/kernel_summary_c2.txt | cut -c1-64,100-170
head -12 $O/kernel_summary_c3s.txt | cut -c1-64,100-170
python - <<PY
import json
d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1])
print('primary', round(d['value']/1e9,3),'G/s', round(d['ms_per_step'],4),'ms/step frac',round(d['roofline']['frac'],3))
s=d.get('secondary'); 
if s: print('secondary', round(s['value']/1e9,3),'G/s', round(s['ms_per_step'],4), 'frac', round(s['roofline']['frac'],3))
c=d.get('cpu_baseline');
if c: print('cpu', c['value'], c.get('steps30_b65536'), c['reference_default_batch'])
for tag in ('c2','c3s'):
    t=json.load(open('$O/pmc_traffic_%s.json' % tag)); print(tag, 'traffic/step GB', t['hbm_bytes_per_step']/1e9, 'B/inter', t['hbm_bytes_per_interaction'], {k:(round(v['read_bytes']/1e9,3), round(v['write_bytes']/1e9,3)) for k,v in t['plan_kernels_total_bytes'].items()})
import os
p='$O/bench_c3_full.json'
if os.path.exists(p):
    f=json.loads(open(p).read().strip().splitlines()[-1]); print('full c3 on one GPU', round(f['value']/1e9,3), round(f['ms_per_step'],4), round(f['roofline']['frac'],3))
PY
