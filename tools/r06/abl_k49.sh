#!/bin/bash
# VERDICT r5 item 1, step one: price the ceiling of a Winograd form of the single-latent step before building it.  The steady-state Euler step launch by launch
# with the K loop of every 3x3 layer cut to 4/9 of its chunks (tools/build_variant.sh abl_k49 -DSF_ABL_K49: timing only, garbage results) against the product build.
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in product abl_k49; do
  rm -rf $R/gpurun_out/trace_$v
  if [ $v = product ]; then unset SF_LIB_PATH; else export SF_LIB_PATH=$R/build_var/$v/libsfnative.so; fi
  SF_PERSIST=0 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/trace_$v -- python3 $R/tools/chainbench.py euler 4 8 > $R/gpurun_out/trace_$v.log 2>&1 || true
  echo "== $v"
  grep "per step" $R/gpurun_out/trace_$v.log | tail -1
  python3 $R/tools/step_trace.py $(ls $R/gpurun_out/trace_$v/*/*kernel_trace.csv | tail -1) 9 tail | head -12
  SF_PERSIST=0 python3 $R/tools/chainbench.py euler 10 30 | tail -1
done
