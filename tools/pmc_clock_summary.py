"""Summarise tools/pmc_clock.sh: per (kernel, grid) clock = GRBM_GUI_ACTIVE / 8 XCDs / duration and
MFMA-pipe utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs).
Usage: python3 tools/pmc_clock_summary.py gpurun_out/pmc_clk"""
import collections
import csv
import glob
import os
import sys


def main():
    d = sys.argv[1]
    cc = sorted(glob.glob(os.path.join(d, "*", "*counter_collection.csv")))[-1]
    kt = sorted(glob.glob(os.path.join(d, "*", "*kernel_trace.csv")))[-1]
    dur = {r["Dispatch_Id"]: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(kt))}
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    meta = {}
    for r in csv.DictReader(open(cc)):
        per[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
        meta[r["Dispatch_Id"]] = (r["Kernel_Name"], r["Grid_Size"] if "Grid_Size" in r else r.get("Grid_Size_X", "?"))
    agg = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0])
    for did, c in per.items():
        name, grid = meta[did]
        if "conv_" not in name or did not in dur:
            continue
        a = agg[(name, grid)]
        a[0] += 1; a[1] += dur[did]; a[2] += c.get("GRBM_GUI_ACTIVE", 0.0); a[3] += c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    for (name, grid), (n, ns, gui, mfma) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        if n == 0 or gui == 0:
            continue
        print(f"{name[:72]:72s} grid {grid:>10} n={n:4d} dur {ns / n / 1e3:8.1f} us clock {gui / 8 / ns:5.2f} GHz mfma_util {mfma / (gui / 8 * 1024):5.2f}")


if __name__ == "__main__":
    main()
