#!/bin/bash
# Stage-A A/B on the GPU box: timing for each kernel variant, parity tests, optional SQ counters.
# Usage: bash scripts/probe_stage_a.sh [tag] [variants...]
TAG=${1:-x}; shift
OUT=gpurun_out/stage_a_$TAG.txt
: > $OUT
VARS=("$@"); [ ${#VARS[@]} -eq 0 ] && VARS=("MST_V2_WPS=2" "MST_V2_WPS=3" "MST_STAGE_A=spw")
for cfg in "${VARS[@]}"; do
  echo "=== $cfg" >> $OUT
  env $cfg timeout -k 10 300 python scripts/probe_melfeat.py 72 >> $OUT 2>&1 || echo "FAILED rc=$?" >> $OUT
done
echo "=== tests (default kernel)" >> $OUT
timeout -k 10 900 python -m pytest tests/test_melfeat_gpu.py -x -q 2>&1 | tail -15 >> $OUT
grep -v "log-mel\|features  " $OUT | tail -40
