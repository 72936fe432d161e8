cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_neumf.py -q -m gpu -x > $O/tests15.txt 2>&1; tail -5 $O/tests15.txt
cd /tmp
for f in 0 1; do
  export DAISY_NMF_FACT=$f
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/pn_$f -o mf -- python $R/tools/neumf_steps.py 2 262144 > $O/neumf_fact$f.txt 2>&1
  tail -2 $O/neumf_fact$f.txt
  python $R/tools/rocprof_summary.py $O/pn_$f 2>/dev/null | head -28 | cut -c1-90,100-170
  rm -rf $O/pn_$f
done
