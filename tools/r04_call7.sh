# round-4 GPU call: A/B of (a) the owners' Adam moments gathered with the rows (default) vs loaded at the commit (apf0),
# (b) nontemporal stores of the item rows in the item pass (qnt) vs default - dev builds (d = 64 only), two passes each
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
L=$PWD/daisyrec_amd/lib
{
for rep in 1 2; do
  for v in dev dev_apf0; do
    echo "== $v"; DAISY_LIB_OVERRIDE=$L/$v/libdaisyrec_hip.so ADAM_MODES=staged timeout 200 python tools/bench_adam.py 2>&1 | grep "Adam"
  done
done
} | tee $O/adam_prefetch_ab.txt
{
for rep in 1 2; do
  for v in dev dev_qnt; do
    for wl in c3s c2; do TAG=$v DAISY_LIB_OVERRIDE=$L/$v/libdaisyrec_hip.so timeout 120 python tools/probe_step.py $wl 40 2>&1 | grep "^\["; done
  done
done
} | tee $O/qnt_ab.txt
