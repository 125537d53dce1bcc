#!/bin/bash
# sweep the direct-kernel knobs on the GPU box (each setting needs its own process)
out=gpurun_out/sweep_direct.txt
: > $out
run() { echo "== $*" >> $out; env "$@" python tools/modbench.py --quick 2>/dev/null | grep -E "dual_cell|infer_state|rollout" >> $out; }
run SF_DIRECT=0
run SF_DIRECT=1
for mt in 1 2 4; do for cpw in 3 5 9; do run SF_DIRECT_MT=$mt SF_DIRECT_CPW=$cpw; done; done
for ks in 1 2 4 8; do run SF_DIRECT_KS=$ks; done
