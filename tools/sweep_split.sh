#!/bin/bash
out=gpurun_out/sweep_split.txt
: > $out
run() { echo "== $*" >> $out; env "$@" python tools/modbench.py --quick 2>/dev/null | grep -E "dual_cell|infer_state|rollout" >> $out; }
run SF_SPLIT=0
for t in 256 384 512 768 1024; do run SF_SPLIT=1 SF_SPLIT_WGS=$t; done
run SF_SPLIT=1 SF_SPLIT_WGS=512 SF_SPLIT_MINCH=1
run SF_SPLIT=1 SF_SPLIT_WGS=512 SF_SPLIT_MINCH=4
