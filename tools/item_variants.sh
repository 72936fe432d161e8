# A/B of development builds of the staged kernels at both shapes, per-kernel times from rocprofv3 (tools/probe_run.sh):
#   bash tools/item_variants.sh <out.txt> <passes> <variant> [variant ...]      variant = a directory under daisyrec_amd/lib
#   (e.g. dev, dev_old, dev_probe1: tools/devlib.sh builds them), optionally variant:ENV=val[,ENV=val]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export ROUND=${ROUND:-r05}
out=$1; passes=$2; shift 2
for pass in $(seq 1 $passes); do
  for wl in c2 c3s; do
    for spec in "$@"; do
      v=${spec%%:*}; envs=""
      if [ "$spec" != "$v" ]; then envs=$(echo "${spec#*:}" | tr ',' ' '); fi
      DAISY_LIB_OVERRIDE=$R/daisyrec_amd/lib/$v/libdaisyrec_hip.so bash $R/tools/probe_run.sh ${v}_${wl}_$pass $wl $envs
    done
  done
done > $out 2>&1
grep -E "^\[|k_staged_(user|item)<|k_part" $out | cut -c1-200
