#!/bin/bash
# first light of the Winograd form of the single-latent 3x3 layers: oracle tests with the form on / off, then the step timing
set -uo pipefail
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for k in 1 0; do
  echo "== SF_WINO_SP=$k"
  SF_WINO_SP=$k timeout 900 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "c64_cells or golden_ode_step or golden_cells or odd_latent or infer_state_of_batched or split_k" 2>&1 | tail -15
done
for k in 1 0; do
  echo "== chain SF_WINO_SP=$k"
  SF_WINO_SP=$k SF_PERSIST=0 timeout 300 python3 tools/chainbench.py euler 10 30 | tail -1
done
cd /tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/trace_wsp
SF_PERSIST=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/trace_wsp -- python3 $GRAFT_REPO_ROOT/tools/chainbench.py euler 4 8 > $GRAFT_REPO_ROOT/gpurun_out/trace_wsp.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/step_trace.py $(ls $GRAFT_REPO_ROOT/gpurun_out/trace_wsp/*/*kernel_trace.csv | tail -1) 9 tail | head -12
