#!/bin/bash
# kernel-by-kernel timeline of one Euler ode_step (rocprofv3 kernel trace of tools/stepbench.py)
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
PER=${PER:-13}
for cfg in "1 50 50" "8 50 50" "1 200 200"; do
  tag=$(echo $cfg | tr ' ' '_')
  rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_$tag -- python3 $R/tools/stepbench.py $cfg 20 > /dev/null 2>$R/gpurun_out/trace_$tag.err
  f=$(find $R/gpurun_out/trace_$tag -name '*kernel_trace.csv' | head -1)
  echo "== $cfg" >> $R/gpurun_out/trace_step.txt
  python3 $R/tools/step_trace.py $f $PER >> $R/gpurun_out/trace_step.txt 2>&1
done
cat $R/gpurun_out/trace_step.txt
