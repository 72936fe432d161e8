# The open plan-build experiment in one GPU call (gpurun --timeout 300 -- 'bash tools/plan_variants.sh'):
# 1. every variant of the epoch-plan build against the default one - all batches compared, ms per build (6 shapes);
# 2. the variants at BASELINE configs[1] and at configs[2] table shapes, timed one after the other in ONE process per
#    workload (same box, twice each), next to the steps' own time: tools/r03_probe.py with PROBE_PLAN_VARIANTS.
# Output: gpurun_out/plan_variants.txt  (copy to profiles/ when it is worth keeping)
cd $GRAFT_REPO_ROOT
O=gpurun_out
mkdir -p $O
{
  timeout 120 python tools/r03_onepass_check.py 2>&1 | grep -E "^n=|ONEPASS"
  for wl in c2 c3s; do
    TAG=plans PROBE_PLAN_VARIANTS=${VARIANTS:-0,0:4096,0:2048,2,3,3:4096,1} timeout 120 python tools/r03_probe.py $wl 40 2>&1 | grep "^\["
  done
} | tee $O/plan_variants.txt
