#!/bin/bash
# SE-scaled / SE-residual 3x3 layers and the sampling layer of the batched infer_state (50x50 images) in the concatenated-image form of
# conv_wino5_kernel (SF_WINO_CAT_SCALED=1) against the plain form (=0: 25 tile columns in 4 blocks of 8): every Winograd launch of a
# batch-32 forward timed by itself (SF_WINO_LIST), then the headline
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/${1:-catsc}
mkdir -p $out
timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_conv_random.py tests/test_gpu_philox.py -x -q -m gpu -p no:cacheprovider -k "infer_state or epilogues_of_the_batched or c64_cells or golden or philox or noise" 2>&1 | tail -4
for v in 0 1; do
  SF_WINO_CAT_SCALED=$v SF_WINO_LIST=1 timeout 600 python3 tools/r05/wino_layers.py 32 2> $out/list_$v.txt > /dev/null
  python3 tools/r05/wino_layers.py --summarise $out/list_$v.txt > $out/summary_$v.txt
  echo "== SF_WINO_CAT_SCALED=$v"; grep -E "in_scale=1|add_scale=1|epi=4" $out/summary_$v.txt
done
for v in 0 1 0 1; do
  SF_WINO_CAT_SCALED=$v timeout 600 python bench.py --steps 10 --warmup 3 --headline-only > $out/bench_$v.json 2> $out/bench_$v.err
  python - <<PY
import json
d=json.loads(open("$out/bench_$v.json").read().strip().splitlines()[-1])
print("bench SF_WINO_CAT_SCALED=$v", round(d["value"],1), round(d["ms_per_step"],2))
PY
done
