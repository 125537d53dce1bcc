#!/bin/bash
# the whole GPU suite again under the opt-in / fallback switches of the library (each a separate pytest process): failures here are bugs of
# paths the default suite reaches only through their own tests.  usage: suite_under_switches.sh [name ...]   (default: all)
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
want() { [ $# -eq 0 ] && return 0; for w in "$@"; do [ "$w" = "$cur" ] && return 0; done; return 1; }
for cur in persist1 wino_sp0 wino16_always wino_sp7_0 ln7_off ln7_always sample_off cat_scaled_off; do
  want "$@" || continue
  case $cur in
    persist1)      run $cur SF_PERSIST=1 ;;
    wino_sp0)      run $cur SF_WINO_SP=0 ;;
    wino16_always) run $cur SF_WINO_SMALL_WGS=1000000000 ;;
    wino_sp7_0)    run $cur SF_WINO_SP7=0 ;;
    ln7_off)       run $cur SF_WINO_LN7=0 ;;
    ln7_always)    run $cur SF_WINO_LN7_MIN_P=0 ;;
    sample_off)    run $cur SF_WINO_SAMPLE=0 ;;
    cat_scaled_off) run $cur SF_WINO_CAT_SCALED=0 ;;
  esac
done
