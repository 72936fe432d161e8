cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { n=$1; shift
  rm -rf /tmp/pmc_$n; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/pmc_$n -o p -- python $R/bench.py --no-cpu-baseline --steps 6 --warmup 2 > /tmp/pmc_$n.log 2>&1
  python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob('/tmp/pmc_$n/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'k_staged_user<' in k or 'k_staged_item<' in k:
            key = 'user' if 'k_staged_user<' in k else 'item'
            agg[key][r['Counter_Name']] += float(r['Counter_Value']); cnt[(key, r['Counter_Name'])] += 1
for key in agg:
    print('$n', key, {c: round(v / cnt[(key, c)]) for c, v in agg[key].items()})
PY
}
run a SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run b SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE
