#!/bin/bash
# kernel trace of a pipelined chain of ODE steps (tools/chainbench.py) -> per-launch timeline of the last steps
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=${1:-chain}
rm -rf $R/gpurun_out/trace_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/trace_$tag -- python3 $R/tools/chainbench.py euler 4 8 > $R/gpurun_out/trace_$tag.log 2>&1
f=$(ls $R/gpurun_out/trace_$tag/*/*kernel_trace.csv | tail -1)
python3 $R/tools/step_trace.py $f ${2:-9} tail > $R/gpurun_out/r03_${tag}_trace.txt
cp $(ls $R/gpurun_out/trace_$tag/*/*kernel_stats.csv | tail -1) $R/gpurun_out/r03_${tag}_kernel_stats.csv
cat $R/gpurun_out/r03_${tag}_trace.txt
