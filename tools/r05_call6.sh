cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -q -m gpu > $O/tests6.txt 2>&1
tail -6 $O/tests6.txt
python tools/adam_small.py > $O/adam_small.txt 2>&1; cat $O/adam_small.txt
for m in default 1; do echo "== DAISY_STAGED_MERGE=$m"; if [ $m = 1 ]; then export DAISY_STAGED_MERGE=1; fi; python tools/sweep_batch.py 16384 32768 65536 131072 262144; done > $O/sweep_merge.txt 2>&1; unset DAISY_STAGED_MERGE; cat $O/sweep_merge.txt
bash tools/pmc_item.sh > $O/counters_after.txt 2>&1; grep "median_us" $O/counters_after.txt
cd /tmp; rocprofv3 --kernel-trace --output-format csv -d /tmp/tg -o t -- python $R/tools/sweep_batch.py 65536 262144 > /dev/null 2>&1; python $R/tools/trace_gaps.py /tmp/tg > $O/trace_gaps_65k.txt 2>&1; tail -40 $O/trace_gaps_65k.txt
