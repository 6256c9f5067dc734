#!/bin/bash
# rocprofv3 kernel stats of one probe script (gpurun): bash scripts/prof_probe.sh TAG scripts/probe_x.py
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
TAG=$1; shift
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_stats -- python3 $R/$1 > $R/gpurun_out/prof_${TAG}_stats.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$R/gpurun_out/prof_${TAG}_stats/*/*_kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print(f"{r['Name'][:80]:80s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:9.1f}")
PY
rm -rf $R/gpurun_out/prof_${TAG}_stats
