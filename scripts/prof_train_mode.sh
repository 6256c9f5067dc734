#!/bin/bash
# rocprofv3 kernel table of one training precision mode.  bash scripts/prof_train_mode.sh <tag> <fp32|f16x3|f16>
set -o pipefail
tag=${1:-r02v}; mode=${2:-f16x3}
out=$PWD/gpurun_out; repo=$PWD
export TRAIN_PRECISION=$mode WARMUP=8 MST_TRAIN_TIMING=1
python scripts/probe_train_step.py > $out/${tag}_probe_$mode.log 2>&1 || exit 1
unset MST_TRAIN_TIMING
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$mode
rocprofv3 --kernel-trace --stats -d /tmp/prof_$mode --output-format csv -- python3 $repo/scripts/probe_train_step.py > $out/${tag}_rocprof_$mode.log 2>&1 || exit 1
cd $repo
python - "$tag" "$mode" <<'PY'
import csv, glob, json, sys
tag, mode = sys.argv[1:3]
print(open(f"gpurun_out/{tag}_probe_{mode}.log").read()[-500:])
fs = glob.glob(f"/tmp/prof_{mode}/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(fs[0])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
ker = [{"name": r["Name"][:110], "calls": int(r["Calls"]), "avg_us": round(float(r["AverageNs"]) / 1e3, 1),
        "pct": round(float(r["Percentage"]), 2)} for r in rows[:24]]
json.dump({"what": f"rocprofv3 --kernel-trace --stats of scripts/probe_train_step.py, TRAIN_PRECISION={mode} (8 warm-up + 5 timed steps of 72 clips)",
           "kernels": ker}, open(f"gpurun_out/{tag}_train_{mode}_kernel_stats.json", "w"), indent=1)
for k in ker[:18]:
    print(f'{k["avg_us"]:9.1f} us x{k["calls"]:4d} {k["pct"]:5.1f}%  {k["name"][:100]}')
PY
