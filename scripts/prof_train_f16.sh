#!/bin/bash
# f16-operand training step: bench lines (f16 / amp / fp32), per-section backward times, rocprofv3 kernel stats.
# Run on the GPU box from the repo root:  bash scripts/prof_train_f16.sh <tag>
set -o pipefail
tag=${1:-r02n}
out=$PWD/gpurun_out
mkdir -p $out
python bench.py --train --train-precision f16 > $out/${tag}_train_f16.json 2> $out/${tag}_err.log || exit 1
python bench.py --train --train-precision amp > $out/${tag}_train_amp.json 2>> $out/${tag}_err.log || exit 1
python bench.py --train > $out/${tag}_train_fp32.json 2>> $out/${tag}_err.log || exit 1
MST_TRAIN_TIMING=1 TRAIN_PRECISION=f16 python scripts/probe_train_step.py > $out/${tag}_probe_f16.log 2>&1 || exit 1
export TRAIN_PRECISION=f16 WARMUP=8
repo=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_f16 --output-format csv -- python3 $repo/scripts/probe_train_step.py > $out/${tag}_rocprof.log 2>&1 || exit 1
cd $repo
python - "$tag" <<'PY'
import csv, glob, json, sys
tag = sys.argv[1]
for f in ("train_f16", "train_amp", "train_fp32"):
    d = json.load(open(f"gpurun_out/{tag}_{f}.json"))
    print(f, d["value"], d["ms_per_step"], d["config"]["loss"], d["config"]["peak_mem_GiB"])
print(open(f"gpurun_out/{tag}_probe_f16.log").read()[-700:])
fs = glob.glob("/tmp/prof_f16/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(fs[0])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
ker = [{"name": r["Name"][:110], "calls": int(r["Calls"]), "avg_us": round(float(r["AverageNs"]) / 1e3, 1),
        "pct": round(float(r["Percentage"]), 2)} for r in rows[:24]]
json.dump({"what": "rocprofv3 --kernel-trace --stats of scripts/probe_train_step.py, TRAIN_PRECISION=f16 (8 warm-up + 5 timed steps of 72 clips)",
           "kernels": ker}, open(f"gpurun_out/{tag}_train_f16_kernel_stats.json", "w"), indent=1)
for k in ker[:16]:
    print(f'{k["avg_us"]:9.1f} us x{k["calls"]:4d} {k["pct"]:5.1f}%  {k["name"][:90]}')
PY
