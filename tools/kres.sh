#!/bin/bash
# Usage: tools/kres.sh <k_unit.hip> [extra flags]: register / scratch / LDS use of every kernel of one translation unit
F=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++20 -O3 -fPIC -ffp-contract=off -fno-slp-vectorize -fno-fast-math -Iinclude -Wno-unused-function \
  --cuda-device-only -c "$F" -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage "$@" 2>&1 | \
  grep -E "Function Name|VGPRs:|VGPR Spill|ScratchSize|SGPR Spill" | sed -E 's/.*remark: [^ ]+ +//; s/ \[-Rpass.*//' | paste - - - - - | \
  sed -E 's/Function Name: _ZN4vimg//'
