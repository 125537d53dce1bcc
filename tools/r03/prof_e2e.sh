#!/bin/bash
# where the 72 ms of config 3 (after the image backbone) go: rocprofv3 kernel stats of tools/e2ebench.py
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_e2e
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_e2e -- python3 $R/tools/e2ebench.py > $R/gpurun_out/prof_e2e.log 2>&1
f=$(ls $R/gpurun_out/prof_e2e/*/*kernel_stats.csv | tail -1)
cp $f $R/gpurun_out/r03_kernel_stats_e2e.csv
python3 - $f <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
calls = 5.0      # fwd() runs 2 warm-ups + 3 timed
print(f"all kernels: {tot / calls / 1e6:.2f} ms per forward")
for r in rows[:16]:
    print(f"{float(r['TotalDurationNs']) / calls / 1e6:8.2f} ms  {int(r['Calls']) / calls:7.1f} calls  {r['Name'][:110]}")
PY
grep -o '"ms_per_sample": [0-9.]*' $R/gpurun_out/prof_e2e.log
