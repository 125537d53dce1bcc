#!/bin/bash
# rocprofv3 kernel stats of the batch-32 forward in the opt-in bf16x3 mode (and fp32 for comparison)
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for m in bf16x3 fp32; do
  rm -rf $R/gpurun_out/prof_$m
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$m -- python3 $R/tools/r03/b3_quick.py ${1:-32} $m > $R/gpurun_out/prof_$m.log 2>&1
  f=$(ls $R/gpurun_out/prof_$m/*/*kernel_stats.csv | tail -1)
  cp $f $R/gpurun_out/r03_kernel_stats_forward_b${1:-32}_$m.csv
  echo "== $m"; head -14 $f | cut -c1-200
done
