# per-kernel time of NeuMF steps: bash tools/dbg/neumf_kstats.sh <level> [B]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/nk; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/nk -o p -- python $R/tools/dbg/neumf_prof.py ${1:-2} ${2:-262144} > /tmp/nk.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/nk/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:24]:
    print(f"{float(r['TotalDurationNs'])/1e3:10.1f} us {int(r['Calls']):5d} calls {float(r['AverageNs'])/1e3:9.1f} us/call {100*float(r['TotalDurationNs'])/tot:5.1f}%  {r['Name'][:110]}")
PY
