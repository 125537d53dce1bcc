#!/bin/bash
# headline workload at other samples-per-forward (information only: the bench line stays at 32)
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/bsweep
for b in 16 32 48 64; do
  timeout 600 python bench.py --steps 6 --warmup 2 --headline-only --batch $b > gpurun_out/bsweep/b$b.json 2> gpurun_out/bsweep/b$b.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/bsweep/b$b.json").read().strip().splitlines()[-1])
print("batch $b", round(d["value"],1), "ODE-steps/s", round(d["ms_per_step"],2), "ms", round(d["ms_per_step"]/$b,3), "ms/sample", round(d["roofline"]["frac"],3))
PY
done
