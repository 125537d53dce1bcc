#!/bin/bash
out=gpurun_out/sweep_dephase.txt
: > $out
for d in 0 50 100 150; do
  echo "== SF_DEPHASE=$d" >> $out
  SF_DEPHASE=$d python tools/modbench.py --tail 2>/dev/null | grep -E "n8 |n16 |n4 256" >> $out
  SF_DEPHASE=$d python tools/modbench.py --bigconvs 2>/dev/null | grep conv >> $out
done
