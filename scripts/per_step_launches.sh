#!/bin/bash
# per-step kernel launch counts of a bench.py mode: two rocprofv3 runs that differ by 10 timed steps (gpurun).  Usage: bash scripts/per_step_launches.sh TAG bench-args...
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
TAG=$1; shift
for n in 4 14; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_n$n -- python3 $R/bench.py --steps $n --warmup 5 --no-cpu-baseline "$@" > $R/gpurun_out/prof_${TAG}_n$n.log 2>&1
done
python3 - <<PY
import csv, glob
def load(n):
    f = glob.glob("$R/gpurun_out/prof_${TAG}_n%d/*/*_kernel_stats.csv" % n)[0]
    return {r["Name"]: (int(r["Calls"]), float(r["TotalDurationNs"])) for r in csv.DictReader(open(f))}
a, b = load(4), load(14)
rows = []
for k, (c, t) in b.items():
    c0, t0 = a.get(k, (0, 0.0))
    if c - c0 > 0:
        rows.append(((c - c0) / 10.0, (t - t0) / 10.0 / 1e3, k))
lib = [r for r in rows if "anonymous namespace" in r[2] or "_GLOBAL__N" in r[2]]
oth = [r for r in rows if r not in lib]
print("per step: libmst launches %.1f (%.2f ms), other launches %.1f (%.3f ms)" % (sum(r[0] for r in lib), sum(r[1] for r in lib) / 1e3, sum(r[0] for r in oth), sum(r[1] for r in oth) / 1e3))
for c, t, k in sorted(oth, key=lambda r: -r[0])[:25]:
    print("  %6.1f x  %8.1f us  %s" % (c, t, k[:110]))
PY
