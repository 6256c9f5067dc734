#!/bin/bash
# Register / spill / LDS figures of the kernels of one csrc/*.hip file for a set of -D flags (cross-compiles, no GPU needed).
# Usage: bash scripts/kernel_resources.sh melfeat.hip "melfeat_v2_kernelIfLi3ELi2ELb1" -DMST_X=1 ...
R=$(cd "$(dirname "$0")/.." && pwd); C=$R/mixing-style-transfer_amd/csrc
F=$1; PAT=$2; shift 2
D=$(mktemp -d)
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-gpu-rdc -Wno-unused-function -I$R/include -I$C "$@" -x hip -c $C/$F --save-temps=obj -o $D/o.o 2>/dev/null
S=$(ls $D/*gfx950.s 2>/dev/null)
if [ -z "$S" ]; then echo "compile failed"; rm -rf $D; exit 1; fi
grep -A12 "\.name:.*$PAT" $S | grep -E "\.name:|sgpr_count|sgpr_spill|vgpr_count|vgpr_spill|private_segment" | sed 's/^ *//' | paste - - - - - - 
cp $S /tmp/last_kernel.s
rm -rf $D
