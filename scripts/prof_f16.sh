#!/bin/bash
# rocprofv3 kernel stats + SQ counters of the eval step in an f16 precision mode (gpurun): bash scripts/prof_f16.sh TAG f16|f16x3-all
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
TAG=${1:-r03f16}
PREC=${2:-f16}
ARGS="--steps 6 --warmup 2 --no-cpu-baseline --precision $PREC"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_stats -- python3 $R/bench.py $ARGS > $R/gpurun_out/prof_${TAG}_stats.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/prof_${TAG}_sq -- python3 $R/bench.py $ARGS > $R/gpurun_out/prof_${TAG}_sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU --output-format csv -d $R/gpurun_out/prof_${TAG}_fetch -- python3 $R/bench.py $ARGS > $R/gpurun_out/prof_${TAG}_sq2.log 2>&1
