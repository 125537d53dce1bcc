#!/bin/bash
# the whole GPU suite again under the opt-in / fallback switches of the library (each a separate pytest process): failures here are bugs of
# paths the default suite reaches only through their own tests
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/switches
run() {
  name=$1; shift
  env "$@" timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider > gpurun_out/switches/$name.log 2>&1
  echo "$name rc=$? $(grep -E 'passed|failed' gpurun_out/switches/$name.log | tail -1)"
  grep -E "^(FAILED|ERROR)" gpurun_out/switches/$name.log | head -20
}
run persist1 SF_PERSIST=1
run wino_sp0 SF_WINO_SP=0
run wino16_always SF_WINO_SMALL_WGS=1000000000
run wino_sp7_0 SF_WINO_SP7=0
