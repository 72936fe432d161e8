# usage: bash tools/probe_run.sh <name> <wl> [ENV=val ...]   -> probe under rocprofv3, kernel summary + the probe line
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${ROUND:-r04}
mkdir -p $O
name=$1; wl=$2; shift 2
for kv in "$@"; do export "$kv"; done
export TAG=$name
rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_$name -o mf -- python $R/tools/probe_step.py $wl 40 > $O/p_$name.log 2>&1
python $R/tools/rocprof_summary.py $O/p_$name 2>/dev/null | grep -E "daisy::k_(staged|unorm|part)" | cut -c1-48,100-170 > $O/p_$name.kern
rm -rf $O/p_$name
grep "^\[" $O/p_$name.log
cat $O/p_$name.kern
