#!/bin/bash
# timing experiments on the f16 training forward of conv2 (MST_CONV2_DBG bits: 1 no MFMA, 2 no weight staging, 4 no raw store)
for mode in f16x3 f16; do
for d in 0 1 2 4 6 7; do
  export MST_CONV2_DBG=$d TRAIN_PRECISION=$mode WARMUP=3
  cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pc2
  rocprofv3 --kernel-trace --stats -d /tmp/pc2 --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/probe_train_step.py > /dev/null 2>&1
  cd $GRAFT_REPO_ROOT
  python - $mode $d <<'PY'
import csv, glob, sys
fs = glob.glob("/tmp/pc2/**/*kernel_stats.csv", recursive=True)
for r in csv.DictReader(open(fs[0])):
    if "conv2_f16x3" in r["Name"]:
        print(sys.argv[1], "dbg", sys.argv[2], round(float(r["AverageNs"]) / 1e3, 1), "us")
PY
done; done
