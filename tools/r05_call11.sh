cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05; mkdir -p $O
for v in dev dev_blk256; do
  export DAISY_LIB_OVERRIDE=$R/daisyrec_amd/lib/$v/libdaisyrec_hip.so
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/pz_$v -o mf -- python $R/bench.py --dist zipf --no-extras --no-secondary --no-cpu-baseline --steps 20 --warmup 5 > $O/zipf_$v.json 2> $O/zipf_$v.err
  python $R/tools/rocprof_summary.py $O/pz_$v 2>/dev/null | grep -E "daisy::k_(staged|unorm|part)" | cut -c1-60,100-170
  rm -rf $O/pz_$v
  python -c "
import json; d=json.loads(open('$O/zipf_$v.json').read().strip().splitlines()[-1]); print('$v', d['ms_per_step'], d['roofline']['frac'], d.get('repeats'))"
done
