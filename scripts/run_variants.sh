#!/bin/bash
# Times scripts/probe_stage_a_layouts.py (or $PROBE) for the default build (also with the run-time A/B switches) and every
# variants/libmst_*.so; on the GPU box.
PROBE=${PROBE:-scripts/probe_stage_a_layouts.py}
OUT=gpurun_out/variants_${1:-x}.txt
: > $OUT
timeout -k 10 200 python $PROBE >> $OUT 2>&1 || echo "default FAILED" >> $OUT
for sw in MST_V2_NOSTD MST_V2_NOPERM; do
  echo -n "$sw=1 " >> $OUT
  env $sw=1 timeout -k 10 200 python $PROBE >> $OUT 2>&1 || echo "$sw FAILED" >> $OUT
done
for so in variants/libmst_*.so; do
  MST_LIB=$PWD/$so timeout -k 10 200 python $PROBE >> $OUT 2>&1 || echo "$so FAILED" >> $OUT
done
timeout -k 10 200 python $PROBE >> $OUT 2>&1
grep -v "amdgpu.ids" $OUT
