#!/bin/bash
# HBM bytes per fused GRU-ODE step: separate FETCH_SIZE / WRITE_SIZE passes over tools/stepbench.py
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
N=20
for cfg in "1 50 50" "8 50 50" "1 200 200"; do
  tag=$(echo $cfg | tr ' ' '_')
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmcs_${c}_$tag -- python3 $R/tools/stepbench.py $cfg $N > /dev/null 2>$R/gpurun_out/pmcs_${c}_$tag.err
  done
done
python3 $R/tools/pmc_step_to_json.py $N
