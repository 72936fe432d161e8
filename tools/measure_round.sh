# round-2 measurement pass (run on the GPU box through gpurun): bench, kernel stats, PMC traffic, other shapes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02
mkdir -p $O
python $R/bench.py > $O/bench_c2.json 2> $O/bench_c2.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o mf -- python $R/bench.py --no-cpu-baseline > $O/prof_bench.log 2>&1
python $R/tools/rocprof_summary.py $O/prof > $O/mf_kernel_summary.txt 2>/dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_rd -o rd -- python $R/bench.py --steps 12 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_wr -o wr -- python $R/bench.py --steps 12 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
python $R/tools/pmc_traffic.py $O/pmc_rd $O/pmc_wr $O/pmc_traffic.json 2097152 "profiles/r02_pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py --steps 12 (round 2 kernels: staged step + partitioned plan)" > /dev/null
python $R/bench.py --workload c3 --no-cpu-baseline > $O/bench_c3_1gpu.json 2> $O/bench_c3.err
python $R/bench.py --no-cpu-baseline --batch 1048576 > $O/bench_c2_b1m.json 2>/dev/null
python $R/bench.py --no-cpu-baseline --batch 65536 > $O/bench_c2_b64k.json 2>/dev/null
python $R/bench.py --no-cpu-baseline --dist zipf > $O/bench_c2_zipf.json 2>/dev/null
python $R/bench.py --no-cpu-baseline --reg 0 > $O/bench_c2_reg0.json 2>/dev/null
python $R/bench.py --no-cpu-baseline --item-mode chunked > $O/bench_c2_chunked.json 2>/dev/null
python $R/bench.py --gpus 2 --backend gloo --workload c2 --batch 2097152 --no-ref > $O/bench_c2_gloo2.json 2> $O/bench_gloo2.err
python $R/tools/sweep_batch.py > $O/batch_sweep.txt 2>/dev/null
python $R/tools/sweep_factors.py > $O/factor_sweep.txt 2>/dev/null
MODE=chunked python $R/tools/sweep_factors.py 64 100 128 200 256 > $O/factor_sweep_chunked.txt 2>/dev/null
python $R/tools/small_epoch.py > $O/small_epoch.txt 2>/dev/null
python $R/tools/bench_neumf.py > $O/bench_neumf.txt 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_neumf -o nmf -- python $R/tools/neumf_steps.py 2 262144 > /dev/null 2>&1
python $R/tools/rocprof_summary.py $O/prof_neumf > $O/neumf_kernel_summary.txt 2>/dev/null
python $R/tools/bench_lightgcn.py > $O/bench_lightgcn.txt 2>/dev/null
cd $R
head -14 $O/mf_kernel_summary.txt | cut -c1-64,100-170
for f in $O/bench_*.json; do echo "$f: $(python -c "
import json,sys
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); print(round(d['value']/1e9,3),'G/s', round(d['ms_per_step'],4),'ms/step frac',round(d['roofline']['frac'],3), d.get('cpu_baseline',{}).get('value'))
except Exception as e: print('ERR',e)
")"; done
python -c "
import json; d=json.load(open('$O/pmc_traffic.json')); print('traffic/step GB', d['hbm_bytes_per_step']/1e9, 'B/inter', d['hbm_bytes_per_interaction']); [print(k, round(v['read_bytes_per_launch']/1e6,1), round(v['write_bytes_per_launch']/1e6,1)) for k,v in d['per_kernel'].items()]"
