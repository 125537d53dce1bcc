#!/bin/bash
# VERDICT r3 item 3, priced instead of built: the steady-state Euler step launch by launch with the small-P kernel's loaders issuing
# NO pixel DMAs (build_variant.sh abl_nopix -DSF_ABL_NO_PIXEL_DMA: timing only, the results are garbage) against the product build.
# What an LDS-resident pixel operand could save at most is the difference per launch.
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in product abl_nopix; do
  rm -rf $R/gpurun_out/trace_abl_$v
  if [ $v = product ]; then unset SF_LIB_PATH; else export SF_LIB_PATH=$R/build_var/abl_nopix/libsfnative.so; fi
  SF_PERSIST=0 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/trace_abl_$v -- python3 $R/tools/chainbench.py euler 4 8 > $R/gpurun_out/trace_abl_$v.log 2>&1
  echo "== $v"
  grep "per step" $R/gpurun_out/trace_abl_$v.log | tail -1
  python3 $R/tools/step_trace.py $(ls $R/gpurun_out/trace_abl_$v/*/*kernel_trace.csv | tail -1) 9 tail | head -10
done
