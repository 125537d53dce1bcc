#!/bin/bash
# kernel-by-kernel timeline of one Euler ode_step for a given "B h w" and launches per step
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cfg="$1"; per=$2
tag=$(echo $cfg | tr ' ' '_')
rm -rf $R/gpurun_out/trace_$tag
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_$tag -- python3 $R/tools/stepbench.py $cfg 20 > /dev/null 2>$R/gpurun_out/trace_$tag.err
f=$(find $R/gpurun_out/trace_$tag -name '*kernel_trace.csv' | head -1)
python3 $R/tools/step_trace.py $f $per
