L=$GRAFT_REPO_ROOT/daisyrec_amd/lib
cd $GRAFT_REPO_ROOT
for wl in c2; do
bash tools/r03_run.sh x1_$wl $wl DAISY_LIB_OVERRIDE=$L/dev_x1/libdaisyrec_hip.so
done
