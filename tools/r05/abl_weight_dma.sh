#!/bin/bash
# VERDICT r4 item 4c (price the weight copies before building anything): the steady-state Euler step launch by launch with the small-P kernel's loaders issuing
# NO weight DMAs (build_variant.sh abl_now -DSF_ABL_NO_WEIGHT_DMA: timing only, the results are garbage) against the product build.
# What weights that need no per-workgroup fetch (one copy per XCD L2 instead of eight, a weight-stationary slice) could save at most is the difference per launch.
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in product abl_now; do
  rm -rf $R/gpurun_out/trace_abl_$v
  if [ $v = product ]; then unset SF_LIB_PATH; else export SF_LIB_PATH=$R/build_var/abl_now/libsfnative.so; fi
  SF_PERSIST=0 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/trace_abl_$v -- python3 $R/tools/chainbench.py euler 4 8 > $R/gpurun_out/trace_abl_$v.log 2>&1
  echo "== $v"
  grep "per step" $R/gpurun_out/trace_abl_$v.log | tail -1
  python3 $R/tools/step_trace.py $(ls $R/gpurun_out/trace_abl_$v/*/*kernel_trace.csv | tail -1) 9 tail | head -10
done
