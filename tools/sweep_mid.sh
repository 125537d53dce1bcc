#!/bin/bash
out=gpurun_out/sweep_mid.txt
: > $out
run() { echo "== $*" >> $out; env "$@" python tools/modbench.py --quick 2>/dev/null | grep -E "rollout" >> $out; }
run SF_MID_TILES=0
run SF_MID_TILES=400
run SF_MID_TILES=640
run SF_MID_TILES=1000
run SF_MID_TILES=640 SF_SPLIT_WGS=768
run SF_MID_TILES=640 SF_SPLIT_WGS=1024
