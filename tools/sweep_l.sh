#!/bin/bash
out=gpurun_out/sweep_l.txt
: > $out
run() { echo "== $*" >> $out; env "$@" python tools/modbench.py --bigconvs 2>/dev/null | grep conv >> $out; }
run SF_L_CFG=0
for c in 5 6 7 8 9; do run SF_L_CFG=$c; done
