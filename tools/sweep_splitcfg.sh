#!/bin/bash
out=gpurun_out/sweep_splitcfg.txt
: > $out
for c in 4 1; do
  echo "== SF_SPLIT_CFG=$c" >> $out
  SF_SPLIT_CFG=$c python tools/modbench.py --quick 2>/dev/null | grep -E "rollout|dual|infer" >> $out
done
SF_SPLIT_CFG=1 python -m pytest tests/test_gpu_forward.py -x -q -m gpu -k "batch or split" 2>&1 | tail -3 >> $out
