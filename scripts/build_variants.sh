#!/bin/bash
# Timing-only ablation builds of the stage-A v2 kernel into variants/ (git-ignored; travels with gpurun).
# Usage: bash scripts/build_variants.sh 1 2 4 ...   (MST_V2_ABLATE bit masks)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/mixing-style-transfer_amd/csrc
mkdir -p $R/variants
# names: plain number = MST_V2_ABLATE mask; ntN = MST_V2_NT=N (cache-hint experiment, full arithmetic)
for a in "$@"; do
  DEF="-DMST_V2_ABLATE=$a"; case $a in nt*) DEF="-DMST_V2_NT=${a#nt}";; vw*) DEF="-DMST_V2_VMWAIT=${a#vw}";; esac
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -Wno-unused-function -I$R/include -I$C $DEF -x hip -c $C/melfeat.hip -o $R/variants/melfeat_ab$a.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $R/variants/libmst_ab$a.so $C/build/aug.hip.o $C/build/encoder.hip.o $C/build/infonce.hip.o $R/variants/melfeat_ab$a.o $C/build/common.cpp.o
  rm $R/variants/melfeat_ab$a.o
done
ls -la $R/variants
