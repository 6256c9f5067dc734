#!/bin/bash
# scripts/probe_enc_stages.py for the default build and every variants/libmst_*.so (PROBE_PREC picks the precision); on the GPU box
OUT=gpurun_out/enc_variants_${1:-x}.txt
: > $OUT
timeout -k 10 200 python scripts/probe_enc_stages.py >> $OUT 2>&1 || echo "default FAILED" >> $OUT
for so in variants/libmst_*.so; do
  MST_LIB=$PWD/$so timeout -k 10 200 python scripts/probe_enc_stages.py >> $OUT 2>&1 || echo "$so FAILED" >> $OUT
done
timeout -k 10 200 python scripts/probe_enc_stages.py >> $OUT 2>&1
grep -v "amdgpu.ids" $OUT
