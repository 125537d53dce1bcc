"""FETCH_SIZE / WRITE_SIZE passes of tools/pmc_step.sh -> profiles/pmc_ode_step.json.
KiB counters, FETCH_SIZE doubled on gfx950 (MI355X_MICROARCH.md, HBM section); all kernels launched by
the N steps are summed (conv + aux), torch's own fill kernels of the set-up excluded by name."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
out = {"commit": __import__("os").environ.get("SF_COMMIT"), "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 tools/stepbench.py B h w %d; "
                 "sum over the sf:: kernels / steps; FETCH_SIZE x2 (gfx950), KiB.  These are L2 <-> fabric bytes (TCC_EA requests; Infinity-Cache hits are counted, MI355X_MICROARCH.md:297): an upper bound on HBM bytes, not HBM bytes" % N, "cases": {}}
for tag in ("1_50_50", "8_50_50", "1_200_200"):
    tot = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        fs = sorted(glob.glob(os.path.join(ROOT, "gpurun_out", f"pmcs_{c}_{tag}", "*", "*counter_collection.csv")))
        if not fs:
            continue
        v, dur = 0.0, 0.0
        for r in csv.DictReader(open(fs[-1])):
            if "sf::" in r["Kernel_Name"] and r["Counter_Name"] == c:
                v += float(r["Counter_Value"])
        tot[c] = v
        kt = fs[-1].replace("counter_collection", "kernel_trace")
        if os.path.exists(kt):
            for r in csv.DictReader(open(kt)):
                if "sf::" in r["Kernel_Name"]:
                    dur += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            tot["kernel_ns_" + c] = dur
    if len(tot) >= 2:
        B = int(tag.split("_")[0])
        by = (2.0 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024.0 / N
        ns = tot.get("kernel_ns_WRITE_SIZE", 0.0) / N
        out["cases"][tag] = {"batch": B, "fabric_bytes_per_step_launch": by, "fabric_bytes_per_sample_step": by / B,
                             "fetch_bytes": 2.0 * tot["FETCH_SIZE"] * 1024.0 / N, "write_bytes": tot["WRITE_SIZE"] * 1024.0 / N,
                             "kernel_us_per_step_under_pmc": ns / 1e3,
                             "fabric_gbs": by / ns if ns else None, "fabric_gbs_over_hbm_peak_8TBs": by / ns / 8000.0 if ns else None}
for d in ("profiles", "gpurun_out"):
    json.dump(out, open(os.path.join(ROOT, d, "pmc_ode_step.json"), "w"), indent=1)
print(json.dumps(out["cases"], indent=1))
