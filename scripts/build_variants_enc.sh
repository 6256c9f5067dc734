#!/bin/bash
# Timing-only ablation builds of encoder.hip into variants/ (git-ignored; travels with gpurun).
# Usage: bash scripts/build_variants_enc.sh MST_F16E_ABLATE=1 MST_F16E_ABLATE=2 ...
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/mixing-style-transfer_amd/csrc
mkdir -p $R/variants
for a in "$@"; do
  n=$(echo $a | tr '=' '_')
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -Wno-unused-function -I$R/include -I$C -D$a -x hip -c $C/encoder.hip -o $R/variants/enc_$n.o &&
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $R/variants/libmst_$n.so $C/build/aug.hip.o $R/variants/enc_$n.o $C/build/infonce.hip.o $C/build/head.hip.o $C/build/melfeat.hip.o $C/build/common.cpp.o &&
    rm $R/variants/enc_$n.o ) &
done
wait
ls -la $R/variants
