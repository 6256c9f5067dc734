#!/bin/bash
# rocprofv3 kernel stats only (gpurun): bash scripts/prof_stats.sh TAG "bench args"
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
TAG=${1:-st}
shift
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_stats -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline "$@" > $R/gpurun_out/prof_${TAG}_stats.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$R/gpurun_out/prof_${TAG}_stats/*/*_kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:9]:
    print(f"{r['Name'][:90]:90s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:9.1f}")
PY
