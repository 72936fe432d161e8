L=$GRAFT_REPO_ROOT/daisyrec_amd/lib
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_plan.py tests/test_gpu_staged.py tests/test_gpu_property.py tests/test_gpu_fit_dist.py tests/test_gpu_lightgcn_dist.py -x -q -m gpu 2>&1 | tail -8
for wl in c2 c3s; do
bash tools/r03_run.sh feis_$wl $wl DAISY_LIB_OVERRIDE=$L/dev/libdaisyrec_hip.so
done
