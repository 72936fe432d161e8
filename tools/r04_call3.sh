# round-4 GPU call: GPU suite, the batch sweep (steps alone through the C epoch loop) with and without a kernel trace,
# the default build at both BASELINE shapes
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/tests3.log 2>&1; echo "pytest rc $?" | tee -a $O/tests3.log
tail -4 $O/tests3.log
timeout 300 python tools/sweep_batch.py 1024 4096 16384 65536 262144 1048576 2097152 2>&1 | grep "^B=" | tee $O/batch_sweep3.txt
for wl in c2 c3s; do ROUND=r04 bash tools/probe_run.sh d3_$wl $wl; done 2>&1 | tee $O/probe_default3.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tg; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tg -o t -- python $GRAFT_REPO_ROOT/tools/sweep_batch.py 1024 4096 16384 65536 262144 > $GRAFT_REPO_ROOT/$O/sweep_traced3.log 2>&1
python $GRAFT_REPO_ROOT/tools/trace_gaps.py /tmp/tg | tee $GRAFT_REPO_ROOT/$O/trace_gaps3.txt
