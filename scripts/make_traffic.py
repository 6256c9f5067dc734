"""profiles/traffic.json from a summarize_prof.py summary: HBM-side bytes per launch of conv1 / conv2 / stage A.
FETCH_SIZE / WRITE_SIZE come from separate rocprofv3 --pmc passes (scripts/prof_r01.sh), in KiB; FETCH_SIZE is doubled on
gfx950 as MI355X_MICROARCH.md prescribes (confirmed by scripts/calib_fetch.hip, profiles/r01_e_calibration.json)."""
import json, sys
summary, passes = sys.argv[1], sys.argv[2]
d = json.load(open(summary))["pmc"]
out = {"passes": passes}
for key, pat in (("conv1", "conv1_resident_kernel"), ("conv2", "conv_kernel<2, 2"), ("stage_a", "melfeat_v2_kernel<float")):
    k = [n for n in d if pat in n]
    if not k:
        continue
    c = d[k[0]]
    f = int(c["FETCH_SIZE"]["mean_per_launch"] * 1024 * 2)
    w = int(c["WRITE_SIZE"]["mean_per_launch"] * 1024)
    out[key] = {"kernel": k[0], "fetch_bytes_corrected": f, "write_bytes": w, "hbm_bytes_per_launch": f + w,
                "note": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (profiles/{passes}_*), 72 clips per launch; "
                        "counters in KiB; FETCH_SIZE x2 per MI355X_MICROARCH.md HBM section (calibrated: profiles/r01_e_calibration.json)"}
print(json.dumps(out, indent=1))
