cd $GRAFT_REPO_ROOT
export DAISY_LIB_OVERRIDE=$GRAFT_REPO_ROOT/daisyrec_amd/lib/dev/libdaisyrec_hip.so
for ps in 0 1 0 1; do TAG=pstream$ps DAISY_STAGED_PSTREAM=$ps timeout 100 python tools/r03_probe.py c3s 40 2>&1 | grep "^\[" ; done
