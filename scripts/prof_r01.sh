#!/bin/bash
# rocprofv3 passes for the contract workload (run on the GPU box through gpurun); summaries land in gpurun_out/.
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
TAG=${1:-r01c}
ARGS="--steps 4 --warmup 2 --no-cpu-baseline ${MST_PROF_ARGS:-}"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_stats -- python3 $R/bench.py $ARGS > $R/gpurun_out/prof_${TAG}_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_${TAG}_fetch -- python3 $R/bench.py $ARGS > $R/gpurun_out/prof_${TAG}_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_${TAG}_write -- python3 $R/bench.py $ARGS > $R/gpurun_out/prof_${TAG}_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/prof_${TAG}_sq -- python3 $R/bench.py $ARGS > $R/gpurun_out/prof_${TAG}_sq.log 2>&1
ls -R $R/gpurun_out/prof_${TAG}_* | head -40
[ -n "${MST_PROF_NO_CALIB:-}" ] && exit 0
# calibration of FETCH_SIZE / WRITE_SIZE for 4/8/16-byte-per-lane streams (1 GiB each)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_${TAG}_calibf -- $R/scripts/calib_fetch.bin > $R/gpurun_out/prof_${TAG}_calibf.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_${TAG}_calibw -- $R/scripts/calib_fetch.bin > $R/gpurun_out/prof_${TAG}_calibw.log 2>&1
