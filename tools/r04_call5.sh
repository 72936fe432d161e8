# round-4 GPU call: per-kernel split of the staged Adam step at 10 M x 1 M shapes, NeuMF step profile (bf16 storage,
# B = 262144) at three split-K slice sizes
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
ADAM_MODES=staged ADAM_SHAPES=2 bash tools/kstats.sh 18 python $R/tools/bench_adam.py 2>&1 | tee $O/adam_kstats.txt
bash tools/kstats.sh 30 python $R/tools/neumf_steps.py 2 262144 2>&1 | tee $O/neumf_kstats.txt
cd $R
for c in 2048 4096 8192 16384; do echo "DAISY_WGRAD_CHUNK=$c"; DAISY_WGRAD_CHUNK=$c timeout 120 python tools/neumf_steps.py 2 262144 2>&1 | tail -2; done | tee $O/neumf_wgrad_chunk.txt
