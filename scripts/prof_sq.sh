#!/bin/bash
# SQ counter passes of a bench.py mode (gpurun): bash scripts/prof_sq.sh TAG bench-args...   -> gpurun_out/prof_TAG_sq{,2}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
TAG=$1; shift
ARGS="--steps 4 --warmup 2 --no-cpu-baseline $@"
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/prof_${TAG}_sq -- python3 $R/bench.py $ARGS > $R/gpurun_out/prof_${TAG}_sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU --output-format csv -d $R/gpurun_out/prof_${TAG}_fetch -- python3 $R/bench.py $ARGS > $R/gpurun_out/prof_${TAG}_sq2.log 2>&1
python3 $R/scripts/summarize_prof.py $TAG > $R/gpurun_out/${TAG}_sum.json
