timeout -k 10 400 python -m pytest tests/test_aug_loss_gpu.py -x -q 2>&1 | tail -3
bash scripts/prof_stats.sh r03aug1 --aug > /dev/null 2>&1
python - <<PY
import csv,glob
f=glob.glob("gpurun_out/prof_r03aug1_stats/*/*_kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:40]:
    n=r["Name"]
    if any(k in n for k in ("aug_","rev_","Cat")):
        print(n[:80].ljust(80), "calls", r["Calls"], "avg_us", round(float(r["AverageNs"])/1e3,1))
PY
tail -1 gpurun_out/prof_r03aug1_stats.log | cut -c1-200
