cd $GRAFT_REPO_ROOT
for lv in 2 0; do timeout 100 python tools/neumf_steps.py $lv 262144 2>&1 | tail -1; done
timeout 100 python tools/neumf_steps.py 0 256 2>&1 | tail -1
timeout 100 python tools/neumf_steps.py 2 65536 2>&1 | tail -1
