#!/bin/bash
# usage: bash tools/devlib.sh <name> ["-DFLAG=1 ..."]  -> daisyrec_amd/lib/dev_<name>/libdaisyrec_hip.so (d=64-only development
# build; only bpr_staged.hip is recompiled with the flags, the other objects are copied from lib/dev)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
D=$R/daisyrec_amd/lib/dev_$1
mkdir -p $D/obj
for o in $R/daisyrec_amd/lib/dev/obj/*.o; do b=$(basename $o); [ "$b" = bpr_staged.o ] || cp -p $o $D/obj/; done
rm -f $D/obj/bpr_staged.o
make -s -C $R/daisyrec_amd/csrc -j8 dev DEV_DIR=../lib/dev_$1 EXTRA="$2" 2>&1 | grep -E "error|warning: unused" || true
ls -la $D/libdaisyrec_hip.so | awk '{print $5, $9}'
