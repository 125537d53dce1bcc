#!/bin/bash
out=gpurun_out/sweep_convs.txt
: > $out
run() { echo "== $*" >> $out; env "$@" python tools/modbench.py --convs 2>/dev/null | grep conv >> $out; }
run SF_SPLIT=0 SF_DIRECT=0
run SF_SPLIT=0 SF_DIRECT=1
run SF_SPLIT=0 SF_DIRECT=1 SF_DIRECT_MT=4
run SF_SPLIT=1 SF_SPLIT_WGS=256
run SF_SPLIT=1 SF_SPLIT_WGS=512
run SF_SPLIT=1 SF_SPLIT_WGS=512 SF_SPLIT_MINCH=1
