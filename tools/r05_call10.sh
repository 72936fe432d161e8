cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
bash tools/measure_round5.sh > gpurun_out/measure5b.txt 2>&1
tail -12 gpurun_out/measure5b.txt
python - <<PY
import json
d = json.loads(open('gpurun_out/r05m/bench.json').read().strip().splitlines()[-1])
for k, v in d.get('extra', {}).items():
    if 'error' in v: print(k, 'ERROR', v['error'][:300]); continue
    print(k, v.get('value'), v.get('unit'), v.get('roofline', {}).get('frac'), v.get('ms_per_step'), v.get('us_per_step'), [(q.get('batch'), q.get('precision'), round(q['ms_per_step'], 3), round(q['roofline']['frac'], 3)) for q in v.get('points', [])])
print('repeats', d.get('repeats'))
PY
