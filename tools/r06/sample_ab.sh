#!/bin/bash
# the sampling layer of the batched infer_state on conv_wino5_kernel (SF_WINO_SAMPLE=1, default) against its direct form (=0)
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/${1:-smp}
mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_conv_random.py tests/test_gpu_philox.py -x -q -m gpu -p no:cacheprovider -k "infer_state or epilogues_of_the_batched or c64 or golden or philox or noise" 2>&1 | tail -6
for v in 0 1; do
  SF_WINO_SAMPLE=$v timeout 600 python bench.py --steps 10 --warmup 3 --headline-only > $out/bench_$v.json 2> $out/bench_$v.err
  python - <<PY
import json
d=json.loads(open("$out/bench_$v.json").read().strip().splitlines()[-1])
print("bench SF_WINO_SAMPLE=$v", round(d["value"],1), round(d["ms_per_step"],2))
for n, k in d["roofline"]["per_kernel"].items():
    if "sample" in n: print("   ", n, k["calls_per_forward"], round(k["ms_per_forward"],3), round(k["tflops"],1), k.get("frac_of_bound"))
PY
done
