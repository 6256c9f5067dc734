#!/bin/bash
# average duration of the kernels whose name contains $2, in one training mode ($1): rocprofv3 over scripts/probe_train_step.py
export TRAIN_PRECISION=$1 WARMUP=8
repo=$PWD
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pkt
rocprofv3 --kernel-trace --stats -d /tmp/pkt --output-format csv -- python3 $repo/scripts/probe_train_step.py > /dev/null 2>&1
cd $repo
python - "$1" "$2" <<'PY'
import csv, glob, sys
fs = glob.glob("/tmp/pkt/**/*kernel_stats.csv", recursive=True)
for r in csv.DictReader(open(fs[0])):
    if sys.argv[2] in r["Name"]:
        print(sys.argv[1], r["Name"][:90], round(float(r["AverageNs"]) / 1e3, 1), "us")
PY
