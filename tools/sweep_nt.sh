#!/bin/bash
out=gpurun_out/sweep_nt.txt
: > $out
for nt in 1 2; do
 for mt in 0 4; do
  echo "== SF_DIRECT_NT=$nt SF_DIRECT_MT=$mt" >> $out
  SF_DIRECT_NT=$nt SF_DIRECT_MT=$mt python tools/modbench.py --quick 2>/dev/null | grep -E "rollout 10|dual|infer" >> $out
  SF_DIRECT_NT=$nt SF_DIRECT_MT=$mt python tools/modbench.py --convs 2>/dev/null | grep "50x50" >> $out
 done
done
SF_DIRECT_NT=2 python -m pytest tests/test_gpu_ops.py tests/test_gpu_forward.py -x -q -m gpu 2>&1 | tail -2 >> $out
