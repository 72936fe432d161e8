#!/bin/bash
# usage: bash tools/kres.sh <file.hip> <mangled-name regex> ["-DFLAG=1 ..."]  -> registers / LDS / scratch / occupancy of the
# matching kernels of a d=64-only build (the compiler's own -Rpass-analysis=kernel-resource-usage remarks)
R=$(cd "$(dirname "$0")/.." && pwd)
cd $R/daisyrec_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -DDAISY_ONLY_D64=1 $3 \
    -Rpass-analysis=kernel-resource-usage -c $1 -o /dev/null 2>&1 |
  awk -v pat="$2" '/Function Name:/ {on = ($0 ~ pat); if (on) {sub(/.*Function Name: /, ""); sub(/ \[-Rpass.*/, ""); printf "%s\n   ", substr($0, 1, 120)}}
       on && /(VGPRs:|AGPRs:|ScratchSize|Occupancy|LDS Size|SGPRs:)/ {sub(/.*remark: [^ ]* /, ""); sub(/ \[-Rpass.*/, ""); printf "%s; ", $0}
       on && /LDS Size/ {printf "\n"}'
