cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -x -q -m gpu > $O/tests5.txt 2>&1
tail -4 $O/tests5.txt | head -3
bash tools/measure_round5.sh > $O/measure5.txt 2>&1
tail -30 $O/measure5.txt
