#!/bin/bash
# 16-tile blocks of conv_wino5_kernel (SF_WINO_SMALL_WGS: launches below that many 32-tile workgroups take them; 0 = never):
# parity of the Winograd-covered tests under both block sizes, then the single-sample forward and the headline per threshold.
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/${1:-w16}
mkdir -p $out
T="tests/test_gpu_ops.py tests/test_gpu_conv_random.py tests/test_gpu_forward.py tests/test_gpu_configs.py tests/test_gpu_bf16x3.py"
for thr in 1000000000 0; do
  SF_WINO_SMALL_WGS=$thr timeout 1500 python -m pytest $T -x -q -m gpu -p no:cacheprovider -k "not both_block_sizes" > $out/tests_$thr.log 2>&1
  echo "tests thr=$thr rc=$? $(tail -1 $out/tests_$thr.log)"
done
for thr in 0 512 700 1000 1400 2600; do
  for b in 1 2; do
    SF_WINO_SMALL_WGS=$thr timeout 300 python tools/r06/batch1_profile.py $b > $out/b${b}_$thr.log 2>&1
    echo "batch=$b thr=$thr $(grep -m1 ms_per_forward $out/b${b}_$thr.log)"
    grep "conv_wino<64x32t2," $out/b${b}_$thr.log
  done
done
for thr in 0 512 2600; do
  SF_WINO_SMALL_WGS=$thr timeout 600 python bench.py --steps 10 --warmup 3 --headline-only > $out/bench_$thr.json 2> $out/bench_$thr.err
  python - <<PY
import json
d=json.loads(open("$out/bench_$thr.json").read().strip().splitlines()[-1])
print("bench thr=$thr", d["value"], d["ms_per_step"], d["roofline"]["frac"])
PY
done
