#!/bin/bash
# usage: bash tools/exp/regtest.sh [extra hipcc flags]  -> registers / scratch of the mutation kernel instantiations in regtest.hip
cd "$(dirname "$0")/../.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Rpass-analysis=kernel-resource-usage --cuda-device-only "$@" -c tools/exp/regtest.hip -o /dev/null 2>&1 |
  grep -E "error|Function Name|VGPRs:|Scratch|Occupancy" | grep -A3 "mutate" | sed -e 's/.*remark: *//' | paste - - - - | sed -e 's/\[-Rpass[^]]*\]//g'
