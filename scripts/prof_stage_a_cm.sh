#!/bin/bash
# SQ counter passes over the stage-A kernel alone (scripts/probe_melfeat_cm.py); run through gpurun.  MST_LIB picks a variant.
# Usage: bash scripts/prof_stage_a_cm.sh TAG
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
TAG=${1:-sa}
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_IFETCH"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/prof_${TAG}_$i -- python3 $R/scripts/probe_melfeat_cm.py > $R/gpurun_out/prof_${TAG}_$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<PY > $R/gpurun_out/prof_${TAG}_summary.txt
import csv, glob, collections
for d in sorted(glob.glob("$R/gpurun_out/prof_${TAG}_[0-9]/")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for row in csv.DictReader(open(f)):
            if "melfeat_v2" in row["Kernel_Name"]:
                a = acc[row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
        for k, (v, n) in acc.items():
            print(f"{k:28s} {v / n:16.0f} per launch ({n} launches)")
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        ts = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(f)) if "melfeat_v2" in r["Kernel_Name"]]
        if ts: print(f"  kernel time under this pass: {sum(ts) / len(ts) / 1e3:.1f} us ({len(ts)} launches)")
PY
cat $R/gpurun_out/prof_${TAG}_summary.txt
rm -rf $R/gpurun_out/prof_${TAG}_[0-9]
