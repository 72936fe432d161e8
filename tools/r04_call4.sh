# round-4 GPU call: GPU suite (row pitch, sparse flavour), factor sweep with the automatic row pitch, per-kernel split of
# the staged Adam step at both shapes, NeuMF step profile (bf16 storage, B = 262144)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/tests4.log 2>&1; echo "pytest rc $?" | tee -a $O/tests4.log
tail -4 $O/tests4.log
timeout 300 python tools/sweep_factors.py 64 50 24 32 100 57 2>&1 | grep "^d=" | tee $O/factor_sweep.txt
PITCH=0 timeout 300 python tools/sweep_factors.py 50 24 57 2>&1 | grep "^d=" | sed 's/^/PITCH=0 /' | tee -a $O/factor_sweep.txt
ADAM_MODES=staged timeout 300 python tools/bench_adam.py 2>&1 | grep "Adam" | tee $O/bench_adam.txt
ADAM_MODES=staged ADAM_SHAPES=2 bash tools/kstats.sh 16 python tools/bench_adam.py 2>&1 | tee $O/adam_kstats.txt
bash tools/kstats.sh 30 python tools/neumf_steps.py 2 262144 2>&1 | tee $O/neumf_kstats.txt
