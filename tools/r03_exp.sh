L=$GRAFT_REPO_ROOT/daisyrec_amd/lib
for wl in c3s c2; do
bash tools/r03_run.sh iblk128_$wl $wl DAISY_LIB_OVERRIDE=$L/dev/libdaisyrec_hip.so DAISY_STAGED_IBLK=128
bash tools/r03_run.sh igrid2k_$wl $wl DAISY_LIB_OVERRIDE=$L/dev/libdaisyrec_hip.so DAISY_STAGED_IGRID=2048
bash tools/r03_run.sh igrid1k_$wl $wl DAISY_LIB_OVERRIDE=$L/dev/libdaisyrec_hip.so DAISY_STAGED_IGRID=1024
done
