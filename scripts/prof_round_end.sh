#!/bin/bash
# end-of-round profiles on the GPU box: contract workload (kernel stats + PMC passes), training kernel tables, per-step launch counts
tag=${1:-r03z}
MST_PROF_NO_CALIB=1 bash scripts/prof_r01.sh $tag > gpurun_out/${tag}_prof_contract.log 2>&1
cd $GRAFT_REPO_ROOT
python scripts/summarize_prof.py $tag > gpurun_out/${tag}_summary.log 2>&1
for m in f16 f16x3 fp32; do bash scripts/prof_train_mode.sh $tag $m > gpurun_out/${tag}_prof_$m.log 2>&1; cd $GRAFT_REPO_ROOT; done
bash scripts/per_step_launches.sh ${tag}ps --train --train-precision f16 > gpurun_out/${tag}_launches.log 2>&1
cd $GRAFT_REPO_ROOT
tail -3 gpurun_out/${tag}_summary.log; head -3 gpurun_out/${tag}_launches.log
