cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/r05/tests17.txt 2>&1; tail -3 gpurun_out/r05/tests17.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 600 python tools/soak.py > gpurun_out/r05/soak.txt 2>&1; tail -6 gpurun_out/r05/soak.txt
