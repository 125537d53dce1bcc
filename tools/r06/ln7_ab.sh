#!/bin/bash
# the batched 7x7 + LayerNorm layer as nine 3x3 tap groups on conv_wino5_kernel (SF_WINO_LN7=1, default) against the direct form (=0)
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/${1:-ln7}
mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -p no:cacheprovider -k "tap_groups or c64" 2>&1 | tail -15
for v in 0 1; do
  SF_WINO_LN7=$v timeout 600 python bench.py --steps 10 --warmup 3 --headline-only > $out/bench_$v.json 2> $out/bench_$v.err
  python - <<PY
import json
d=json.loads(open("$out/bench_$v.json").read().strip().splitlines()[-1])
print("bench SF_WINO_LN7=$v", d["value"], d["ms_per_step"], d["roofline"]["frac"])
for n, k in d["roofline"]["per_kernel"].items():
    if "ln_gelu" in n or "trust" in n: print("   ", n, k["calls_per_forward"], round(k["ms_per_forward"],3), round(k["tflops"],1), k.get("frac_of_bound"))
PY
done
