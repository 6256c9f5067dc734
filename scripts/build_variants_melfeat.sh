#!/bin/bash
# Timing / A-B builds of melfeat.hip into variants/ (git-ignored; travels with gpurun).
# Usage: bash scripts/build_variants_melfeat.sh MST_X=1 "MST_X=1 -DMST_Y=2" ...
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/mixing-style-transfer_amd/csrc
mkdir -p $R/variants
for a in "$@"; do
  n=$(echo $a | tr '= ' '__' | tr -d '-')
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -Wno-unused-function -I$R/include -I$C -D$a -x hip -c $C/melfeat.hip -o $R/variants/mf_$n.o &&
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $R/variants/libmst_$n.so $C/build/aug.hip.o $C/build/encoder.hip.o $C/build/infonce.hip.o $C/build/head.hip.o $R/variants/mf_$n.o $C/build/common.cpp.o &&
    rm $R/variants/mf_$n.o ) &
done
wait
ls -la $R/variants
