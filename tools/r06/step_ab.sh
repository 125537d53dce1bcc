#!/bin/bash
# same-box A/B of the single-latent step under environment switches: steady-state step time + stamps of two launches (diagnostic build present)
# usage (through gpurun): AB_VARIANTS="SF_SP_SHORT_TAIL=1 SF_SP_SHORT_TAIL=0 SF_WINO_SP7=0" bash tools/r06/step_ab.sh
set -uo pipefail
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for v in ${AB_VARIANTS:-SF_SP_SHORT_TAIL=1 SF_SP_SHORT_TAIL=0 SF_WINO_SP7=0}; do
  echo "== $v"
  env $v SF_PERSIST=0 timeout 300 python3 tools/chainbench.py euler 10 30 2>/dev/null | tail -1
  env $v SF_LIB_PATH=build_var/stamp/libsfnative.so SF_PERSIST=0 timeout 300 python3 tools/r06/stamps_rollout.py 3 2>/dev/null | sed -n 12,13p | cut -c 1-200
done
