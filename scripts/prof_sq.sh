#!/bin/bash
# SQ counter passes (separate --pmc runs) over the kernels of one probe whose names contain FILTER; run through gpurun.
# Usage: [PROBE_PREC=...] bash scripts/prof_sq.sh TAG FILTER scripts/probe_x.py
# GRBM_GUI_ACTIVE is summed over the 8 XCDs: / 8 / kernel time = the clock the chip held during the kernel.
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
TAG=$1; FILTER=$2; PROBE=$3
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_BUSY_CU_CYCLES"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/prof_${TAG}_$i -- python3 $R/$PROBE > $R/gpurun_out/prof_${TAG}_$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<PY > $R/gpurun_out/prof_${TAG}_summary.txt
import csv, glob, collections
for d in sorted(glob.glob("$R/gpurun_out/prof_${TAG}_[0-9]/")):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "$FILTER" in row["Kernel_Name"]:
                a = acc[(row["Kernel_Name"][:48], row["Counter_Name"])]; a[0] += float(row["Counter_Value"]); a[1] += 1
    for (kn, k), (v, n) in sorted(acc.items()):
        print(f"{kn:48s} {k:28s} {v / n:16.0f} per launch ({n} launches)")
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        ts = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "$FILTER" in r["Kernel_Name"]: ts[r["Kernel_Name"][:48]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        for kn, t in ts.items(): print(f"  {kn}: kernel time under this pass {sum(t) / len(t) / 1e3:.1f} us ({len(t)} launches)")
PY
cat $R/gpurun_out/prof_${TAG}_summary.txt
rm -rf $R/gpurun_out/prof_${TAG}_[0-9]
