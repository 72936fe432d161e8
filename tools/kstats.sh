# per-kernel time of a command: bash tools/kstats.sh <n rows> <cmd...>
cd /tmp && export TMPDIR=/tmp
N=$1; shift
rm -rf /tmp/ks; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -o p -- "$@" > /tmp/ks.log 2>&1
tail -2 /tmp/ks.log | cut -c1-600
python - $N <<'PY'
import csv, glob, sys
f = glob.glob('/tmp/ks/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f"total kernel time {tot/1e6:.3f} ms")
for r in rows[:int(sys.argv[1])]:
    print(f"{float(r['TotalDurationNs'])/1e3:10.1f} us {int(r['Calls']):5d} calls {float(r['AverageNs'])/1e3:9.2f} us/call {100*float(r['TotalDurationNs'])/tot:5.1f}%  {r['Name'][:100]}")
PY
