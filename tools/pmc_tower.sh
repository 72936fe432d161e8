cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { n=$1; shift
  rm -rf /tmp/pmc_$n; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/pmc_$n -o p -- python $R/tools/neumf_steps.py 2 262144 > /tmp/pmc_$n.log 2>&1
  python - <<PY
import csv, glob, collections
agg = collections.defaultdict(float); cnt = collections.Counter()
for f in glob.glob('/tmp/pmc_$n/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_nmf_tower<' in r['Kernel_Name']:
            agg[r['Counter_Name']] += float(r['Counter_Value']); cnt[r['Counter_Name']] += 1
print('$n', {c: round(v / cnt[c]) for c, v in agg.items()})
PY
}
run a SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run b SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MFMA SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE
run c SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM
