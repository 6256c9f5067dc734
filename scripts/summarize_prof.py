"""Summarise rocprofv3 output dirs (kernel stats + PMC passes) into small CSV/JSON files for profiles/."""
import csv, glob, json, os, sys, collections
tag = sys.argv[1]
base = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
out = {}
def short(n):
    n = n.replace("(anonymous namespace)::", "")
    return n.split("(")[0][:70]
f = glob.glob(f"{base}/prof_{tag}_stats/*/*_kernel_stats.csv")
if f:
    rows = list(csv.DictReader(open(f[0])))
    out["kernel_stats"] = [{"name": short(r["Name"]), "calls": int(r["Calls"]), "avg_us": round(float(r["AverageNs"]) / 1e3, 2),
                            "pct": float(r["Percentage"])} for r in rows[:24]]
for pass_ in ("fetch", "write", "sq"):
    f = glob.glob(f"{base}/prof_{tag}_{pass_}/*/*_counter_collection.csv")
    if not f:
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        if not any(x in k for x in ("melfeat", "conv", "attn_", "film_", "proj_", "read_k", "write_k", "aug_", "rev_")):
            continue
        for c, v in cs.items():
            out.setdefault("pmc", {}).setdefault(k, {})[c] = {"mean_per_launch": sum(v) / len(v), "launches": len(v)}
print(json.dumps(out, indent=1))
