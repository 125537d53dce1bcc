#!/bin/bash
# HBM bytes per launch of the lift-splat pooling kernels: separate FETCH_SIZE / WRITE_SIZE passes
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmcl_$c -- python3 $R/tools/liftbench.py --reps 5 > /dev/null 2>$R/gpurun_out/pmcl_$c.err
done
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/lift_trace -- python3 $R/tools/liftbench.py --reps 5 > /dev/null 2>$R/gpurun_out/lift_trace.err
python3 - <<'PY'
import csv, glob, json, os, collections
R = os.environ["GRAFT_REPO_ROOT"]
agg = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = sorted(glob.glob(os.path.join(R, "gpurun_out", "pmcl_" + c, "*", "*counter_collection.csv")))[-1]
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == c and ("lift_" in r["Kernel_Name"] or "radix" in r["Kernel_Name"] or "onesweep" in r["Kernel_Name"]
                                       or "depth_softmax" in r["Kernel_Name"]):
            d[r["Kernel_Name"][:90]].append(float(r["Counter_Value"]))
    agg[c] = d
out = {"method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 tools/liftbench.py --reps 5; "
                 "KiB counters, FETCH_SIZE x2 (gfx950), average per launch", "kernels": {}}
for k in agg["FETCH_SIZE"]:
    f = agg["FETCH_SIZE"][k]; w = agg["WRITE_SIZE"].get(k, [0.0])
    out["kernels"][k] = {"launches": len(f), "fetch_bytes": 2048.0 * sum(f) / len(f), "write_bytes": 1024.0 * sum(w) / len(w),
                         "hbm_bytes_per_launch": 2048.0 * sum(f) / len(f) + 1024.0 * sum(w) / len(w)}
dom = [k for k in out["kernels"] if "lift_pool_kernel<0>" in k]
if dom:
    out["hbm_bytes_per_launch"] = out["kernels"][dom[0]]["hbm_bytes_per_launch"]
    out["kernel"] = dom[0]
for d in ("profiles", "gpurun_out"):
    json.dump(out, open(os.path.join(R, d, "pmc_lift_pool.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
PY
