cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05; mkdir -p $O
bash $R/tools/item_variants.sh $O/ab9c.txt 2 dev_nopark dev
cd $R
timeout 900 python -m pytest tests/test_gpu_staged.py tests/test_gpu_plan.py tests/test_gpu_fit_dist.py tests/test_gpu_property.py -q -m gpu -x > $O/tests9.txt 2>&1; tail -3 $O/tests9.txt
