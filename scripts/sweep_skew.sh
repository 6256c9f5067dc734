#!/bin/bash
# conv1 / conv2 of the contract forward by start-up skew (MST_CONV1_SKEW / MST_CONV2_SKEW, units of 1024 cycles; + 256 = odd waves)
OUT=gpurun_out/skew_${1:-x}.txt
: > $OUT
for prec in fp32 f16x3-all f16; do
  for sk in ${SKEWS:-0 2 4 8 16 260 264}; do
    PROBE_PREC=$prec MST_CONV1_SKEW=$sk MST_CONV2_SKEW=$sk timeout -k 10 120 python scripts/probe_enc_stages.py >> $OUT 2>&1 || echo "$prec $sk FAILED" >> $OUT
  done
done
grep -v "amdgpu.ids" $OUT
