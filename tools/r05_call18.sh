cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05; mkdir -p $O
for wl in c2 c3s; do
  DAISY_LIB_OVERRIDE=$R/daisyrec_amd/lib/dev_planx/libdaisyrec_hip.so PROBE_SIDECOUNT=1 TAG=sidecount python $R/tools/probe_step.py $wl 20
done > $O/sidecount.txt 2>&1
grep "^\[" $O/sidecount.txt
cd $R; timeout 600 python -m pytest tests/test_gpu_staged.py -q -m gpu -x -k "slices" 2>&1 | tail -2
