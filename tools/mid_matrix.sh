cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05; mkdir -p $O
for ug in 512 1024 2048; do for mg in 0 1; do
  echo "== UGRID=$ug MERGE=$mg"; DAISY_STAGED_UGRID=$ug DAISY_STAGED_MERGE=$mg python $R/tools/sweep_batch.py 16384 32768 65536 131072 2>&1 | grep "B="
done; done > $O/mid_matrix.txt 2>&1
cat $O/mid_matrix.txt
