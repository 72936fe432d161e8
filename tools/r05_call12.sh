cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_staged.py tests/test_gpu_plan.py tests/test_gpu_dist.py tests/test_gpu_fit_dist.py -q -m gpu -x > $O/tests12.txt 2>&1; tail -4 $O/tests12.txt
cd /tmp
for eb in 0 auto; do
  if [ $eb = 0 ]; then export DAISY_EDGE_BLOCKS=0; else unset DAISY_EDGE_BLOCKS; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/pz_$eb -o mf -- python $R/bench.py --dist zipf --no-extras --no-secondary --no-cpu-baseline --steps 20 --warmup 5 > $O/zipf_eb$eb.json 2> $O/zipf_eb$eb.err
  python $R/tools/rocprof_summary.py $O/pz_$eb 2>/dev/null | grep -E "daisy::k_staged" | cut -c1-60,100-170
  rm -rf $O/pz_$eb
  python -c "
import json; d=json.loads(open('$O/zipf_eb$eb.json').read().strip().splitlines()[-1]); print('zipf eb=$eb', d['ms_per_step'], d['roofline']['frac'], d.get('repeats'))"
done
unset DAISY_EDGE_BLOCKS
python $R/bench.py --no-extras --no-secondary --no-cpu-baseline --steps 20 --warmup 5 | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('uniform', d['ms_per_step'], d['roofline']['frac'], d.get('repeats'))"
