import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "melfeat_kernel" in r["Kernel_Name"]]
i0 = idx[-2]
t0 = int(rows[i0]["Start_Timestamp"])
prev_end = t0
for r in rows[i0:idx[-1] + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "")[:60]
    print(f"{(s - t0) / 1e3:10.1f} us  dur {(e - s) / 1e3:9.1f}  gap {(s - prev_end) / 1e3:8.1f}  {name}")
    prev_end = e
