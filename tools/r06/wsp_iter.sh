#!/bin/bash
# one iteration on the Winograd form of the single-latent step: oracle tests, steady-state step time (form on / off), in-kernel stamps (diagnostic build)
set -uo pipefail
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "c64_cells or golden_ode_step or golden_cells or odd_latent or infer_state_of_batched or split_k" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_persistent.py tests/test_gpu_configs.py -q -m gpu -x -k "c64_stream40_rollout_vs_oracle or stream40_hipgraph_equals_eager" 2>&1 | tail -3
for k in 1 0 1; do
  echo "== chain SF_WINO_SP=$k"
  SF_WINO_SP=$k SF_PERSIST=0 timeout 300 python3 tools/chainbench.py euler 10 30 2>/dev/null | tail -1
done
for v in ${WSP_VARIANTS:-SF_WINO_SP7=0}; do
  echo "== chain $v"
  env $v SF_PERSIST=0 timeout 300 python3 tools/chainbench.py euler 10 30 2>/dev/null | tail -1
done
for v in ${STAMP_VARIANTS:-stamp}; do
  if [ -f build_var/$v/libsfnative.so ]; then
    echo "== stamps $v"
    SF_LIB_PATH=build_var/$v/libsfnative.so SF_PERSIST=0 timeout 300 python3 tools/r06/stamps_rollout.py 3 2>/dev/null | sed -n 10,19p
  fi
done
