# round-4 GPU call: GPU suite with the fused user-side Adam catch-up, Adam bench + kernel split
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/tests6.log 2>&1; echo "pytest rc $?" | tee -a $O/tests6.log
tail -4 $O/tests6.log
ADAM_MODES=staged timeout 300 python tools/bench_adam.py 2>&1 | grep "Adam" | tee $O/bench_adam6.txt
ADAM_MODES=staged ADAM_SHAPES=2 bash tools/kstats.sh 12 python $R/tools/bench_adam.py 2>&1 | grep -v "^W2026" | tee $O/adam_kstats6.txt
