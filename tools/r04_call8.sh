# round-4 GPU call: GPU suite with the three-launch step, batch sweep with DAISY_STAGED_MERGE 0 / auto (same box), trace
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/tests8.log 2>&1; echo "pytest rc $?" | tee -a $O/tests8.log
tail -4 $O/tests8.log
for rep in 1 2; do
  for m in 0 auto; do
    if [ $m = auto ]; then unset DAISY_STAGED_MERGE; else export DAISY_STAGED_MERGE=$m; fi
    timeout 200 python tools/sweep_batch.py 1024 4096 16384 65536 2>&1 | grep "^B=" | sed "s/^/merge=$m /"
  done
done | tee $O/batch_sweep8.txt
unset DAISY_STAGED_MERGE
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tg; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tg -o t -- python $GRAFT_REPO_ROOT/tools/sweep_batch.py 1024 4096 16384 > $GRAFT_REPO_ROOT/$O/sweep_traced8.log 2>&1
python $GRAFT_REPO_ROOT/tools/trace_gaps.py /tmp/tg | tee $GRAFT_REPO_ROOT/$O/trace_gaps8.txt
