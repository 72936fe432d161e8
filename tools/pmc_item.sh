# counter passes of k_staged_item / k_staged_user at both shapes: wave-cycle split (the guide:
# SQ_WAIT_ANY + SQ_WAIT_INST_ANY + SQ_ACTIVE_INST_ANY ~ SQ_WAVE_CYCLES), L2 hits / misses, fabric read requests, instruction
# mix.  usage (on the GPU box): bash tools/pmc_item.sh > gpurun_out/r05/counters.txt     [DAISY_LIB_OVERRIDE=... for a dev build]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
pmc() { n=$1; wl=$2; shift 2
  rm -rf /tmp/pmc_$n; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/pmc_$n -o p -- python $R/tools/probe_step.py $wl 12 > /tmp/pmc_$n.log 2>&1
  python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
dur = collections.defaultdict(list)
for f in glob.glob('/tmp/pmc_$n/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        key = 'user' if 'k_staged_user<' in k else ('item' if 'k_staged_item<' in k else None)
        if key is None: continue
        agg[key][r['Counter_Name']] += float(r['Counter_Value']); cnt[(key, r['Counter_Name'])] += 1
for f in glob.glob('/tmp/pmc_$n/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        key = 'user' if 'k_staged_user<' in k else ('item' if 'k_staged_item<' in k else None)
        if key: dur[key].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for key in sorted(agg):
    d = sorted(dur[key]); med = d[len(d) // 2] / 1e3 if d else 0
    print('$n $wl', key, 'median_us_under_pmc', round(med, 1), {c: round(v / cnt[(key, c)]) for c, v in agg[key].items()})
PY
}
for wl in c2 c3s; do
  pmc a_$wl $wl SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE
  pmc b_$wl $wl TCC_HIT_sum TCC_MISS_sum
  pmc c_$wl $wl TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
  pmc d_$wl $wl SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_INST_LEVEL_VMEM
done
