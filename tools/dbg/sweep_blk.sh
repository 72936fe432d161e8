cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "256 256" "128 256" "256 128" "128 128"; do set -- $cfg
  rm -rf /tmp/pu; DAISY_STAGED_UBLK=$1 DAISY_STAGED_IBLK=$2 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pu -- python $R/bench.py --no-cpu-baseline --steps 10 --warmup 2 > /tmp/pu.log 2>&1
  echo "UBLK=$1 IBLK=$2 user/item total ms over 12 calls: $(python $R/tools/rocprof_summary.py /tmp/pu | grep -E "k_staged_user<|k_staged_item<|staged_user_edges|staged_item_edges" | awk '{print $(NF-4)}' | paste -sd' ')  $(tail -1 /tmp/pu.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4))")"
done
