#!/bin/bash
# A/B builds of aug.hip into variants/ (git-ignored; travels with gpurun).  Usage: bash scripts/build_variants_aug.sh "MST_AUG_FT=512 -DMST_AUG_FL=64 ..." ...
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/mixing-style-transfer_amd/csrc
mkdir -p $R/variants
for a in "$@"; do
  n=$(echo $a | tr '= ' '__' | tr -d '-' | sed 's/DMST_AUG_//g; s/MST_AUG_//g')
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -Wno-unused-function -I$R/include -I$C -D$a -x hip -c $C/aug.hip -o $R/variants/aug_$n.o &&
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $R/variants/libmst_aug_$n.so $R/variants/aug_$n.o $C/build/encoder.hip.o $C/build/infonce.hip.o $C/build/head.hip.o $C/build/melfeat.hip.o $C/build/common.cpp.o &&
    rm $R/variants/aug_$n.o ) &
done
wait
ls $R/variants
