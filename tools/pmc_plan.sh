# counter passes of the plan-build kernels (k_part_count / k_part_scatter) at BASELINE configs[1]: what binds them?
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
pmc() { n=$1; shift
  rm -rf /tmp/pmcp_$n; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/pmcp_$n -o p -- python $R/tools/prof_plan.py > /tmp/pmcp_$n.log 2>&1
  python - <<PY
import csv, glob, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); dur = collections.defaultdict(list)
def key_of(k):
    m = re.search(r'k_part_(count|scatter)<([^>]*)>', k)
    return m.group(0) if m else None
for f in glob.glob('/tmp/pmcp_$n/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        key = key_of(r['Kernel_Name'])
        if key: agg[key][r['Counter_Name']] += float(r['Counter_Value']); cnt[(key, r['Counter_Name'])] += 1
for f in glob.glob('/tmp/pmcp_$n/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        key = key_of(r['Kernel_Name'])
        if key: dur[key].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for key in sorted(agg):
    d = sorted(dur[key]); med = d[len(d) // 2] / 1e3 if d else 0
    print('$n', key, 'median_us_under_pmc', round(med, 1), {c: round(v / cnt[(key, c)]) for c, v in agg[key].items()})
PY
}
pmc a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU
pmc b SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR
pmc c SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM
