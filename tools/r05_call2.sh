# round-5 GPU call 2: A/B of the item / user pass changes (no store-wait stalls in the commits, finisher chains, parked rows)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export ROUND=r05
O=$R/gpurun_out/r05
mkdir -p $O
DEV=$R/daisyrec_amd/lib
for pass in 1 2; do
for wl in c2 c3s; do
  for v in dev_old dev dev_notouch; do
    DAISY_LIB_OVERRIDE=$DEV/$v/libdaisyrec_hip.so bash $R/tools/probe_run.sh ${v}_${wl}_$pass $wl
  done
done
done > $O/ab2.txt 2>&1
for v in dev_old dev; do echo "== $v"; DAISY_LIB_OVERRIDE=$DEV/$v/libdaisyrec_hip.so python $R/tools/sweep_batch.py 1024 4096 16384 65536 262144 1048576; done > $O/sweep2.txt 2>&1
cd $R
timeout 900 python -m pytest tests/test_gpu_staged.py tests/test_gpu_parity.py tests/test_gpu_hardening.py tests/test_gpu_fm.py tests/test_gpu_plan.py tests/test_gpu_property.py -x -q -m gpu > $O/tests2.txt 2>&1
grep -E "^\[|k_staged_(user|item)<" $O/ab2.txt | cut -c1-200; cat $O/sweep2.txt; tail -5 $O/tests2.txt
