cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { # name counters...
  n=$1; shift
  timeout 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/pmc_$n -o p -- python $R/tools/dbg/steps.py 2 4 > $R/gpurun_out/pmc_$n.log 2>&1
}
run a SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run b SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE
ls $R/gpurun_out/pmc_a $R/gpurun_out/pmc_b
tail -3 $R/gpurun_out/pmc_a.log $R/gpurun_out/pmc_b.log
