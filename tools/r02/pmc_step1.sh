#!/bin/bash
# HBM / fabric bytes of the single-latent GRU-ODE step only (FETCH_SIZE, WRITE_SIZE in separate passes); prints MB per step
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
N=20
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc1_$c
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc1_$c -- python3 $R/tools/stepbench.py 1 50 50 $N > /dev/null 2>$R/gpurun_out/pmc1_$c.err
done
python3 - <<PY
import csv, glob
tot = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = sorted(glob.glob("$R/gpurun_out/pmc1_%s/*/*counter_collection.csv" % c))[-1]
    tot[c] = sum(float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "sf::" in r["Kernel_Name"] and r["Counter_Name"] == c)
print("fetch %.1f MB  write %.1f MB  total %.1f MB per step" % (2 * tot["FETCH_SIZE"] * 1024 / $N / 1e6, tot["WRITE_SIZE"] * 1024 / $N / 1e6, (2 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024 / $N / 1e6))
PY
