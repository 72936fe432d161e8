cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_staged.py tests/test_gpu_hardening.py tests/test_gpu_parity.py tests/test_gpu_fit_dist.py tests/test_gpu_fm.py -q -m gpu -x -k "adam or Adam or kat" > $O/tests14.txt 2>&1; tail -3 $O/tests14.txt
cd /tmp
ADAM_MODES=staged rocprofv3 --kernel-trace --stats --output-format csv -d $O/pa -o mf -- python $R/tools/bench_adam.py > $O/bench_adam.txt 2>&1
cat $O/bench_adam.txt | grep Adam
python $R/tools/rocprof_summary.py $O/pa 2>/dev/null | grep -E "daisy::k_(staged|adam|unorm)" | cut -c1-70,100-170
rm -rf $O/pa
