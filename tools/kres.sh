#!/bin/bash
# usage: tools/kres.sh file.hip [filter]  -- per-kernel VGPR / scratch / LDS / occupancy from hipcc's resource-usage remarks
f=$1; filt=${2:-.}
d=/tmp/kres_$$; mkdir -p $d
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -c $f -o $d/o.o -save-temps=obj -Rpass-analysis=kernel-resource-usage ${KFLAGS} 2>&1 | \
  grep -E "Function Name|VGPRs:|ScratchSize|LDS Size|Occupancy" | sed 's/\[-Rpass.*//; s/.*remark: [^:]*:[0-9]*:[0-9]*: //' | paste - - - - - | sed 's/ \+/ /g' | grep -E "$filt"
echo "asm: $d/$(basename ${f%.hip})-hip-amdgcn-amd-amdhsa-gfx950.s"
