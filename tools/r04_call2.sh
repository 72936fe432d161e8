# round-4 GPU call: the whole GPU suite, then the AoS plan + item-pass variants at both shapes, then where launch-bound
# steps spend their time.  Output under gpurun_out/r04.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "pytest rc $?" | tee -a $O/tests.log
tail -5 $O/tests.log
for wl in c2 c3s; do ROUND=r04 bash tools/probe_run.sh default_$wl $wl; done 2>&1 | tee $O/probe_default.txt
{
for rep in 1 2; do
  for v in base cw3 ke8; do
    for wl in c2 c3s; do
      TAG=$v DAISY_LIB_OVERRIDE=$PWD/daisyrec_amd/lib/dev_$v/libdaisyrec_hip.so timeout 120 python tools/probe_step.py $wl 40 2>&1 | grep "^\["
    done
  done
  for wl in c2 c3s; do TAG=dev_default DAISY_LIB_OVERRIDE=$PWD/daisyrec_amd/lib/dev/libdaisyrec_hip.so timeout 120 python tools/probe_step.py $wl 40 2>&1 | grep "^\["; done
done
} | tee $O/variants.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tg; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tg -o t -- python $GRAFT_REPO_ROOT/tools/sweep_batch.py 1024 4096 16384 65536 262144 > $GRAFT_REPO_ROOT/$O/sweep_traced.log 2>&1
python $GRAFT_REPO_ROOT/tools/trace_gaps.py /tmp/tg | tee $GRAFT_REPO_ROOT/$O/trace_gaps.txt
cd $GRAFT_REPO_ROOT
timeout 200 python tools/sweep_batch.py 1024 4096 16384 65536 262144 1048576 2097152 2>&1 | tee $O/batch_sweep.txt
