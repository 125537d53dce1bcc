#!/bin/bash
out=gpurun_out/sweep_l.txt
: > $out
run() { echo "== $*" >> $out; env "$@" python tools/modbench.py --bigconvs 2>/dev/null | grep conv >> $out; }
run SF_L_CFG=23
run SF_L_CFG=15
