#!/bin/bash
out=gpurun_out/sweep_fast.txt
: > $out
for d in 0 1; do
  echo "== SF_FAST_SRC=$d" >> $out
  SF_FAST_SRC=$d python tools/modbench.py --bigconvs 2>/dev/null | grep conv >> $out
done
python -m pytest tests/test_gpu_ops.py tests/test_gpu_forward.py -x -q -m gpu 2>&1 | tail -3 >> $out
