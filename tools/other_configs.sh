#!/bin/bash
# bench.py at the other BASELINE configs (4: 16 future steps, 5: 0.05 s x 40 streaming steps with euler / midpoint / rk4,
# 1: C=32 four Euler steps) -> gpurun_out/other_configs.txt: args, ODE-steps/s, ms per forward, ODE steps per sample, ms per sample
out=gpurun_out/other_configs.txt
: > $out
run() { python bench.py --no-cpu-baseline --no-extras --no-roofline --steps 5 --warmup 2 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(' '.join(sys.argv[1:]), round(d['value'],1), round(d['ms_per_step'],2), d['config']['workload'].split('euler: ')[-1].split('rk4: ')[-1].split('midpoint: ')[-1][:40], round(d['ms_per_sample'],2), 'single-sample fwd ms', round(d['single_sample_forward_ms'],2))" "$@" >> $out; }
run --timeset future16 --batch 4
run --timeset future16 --batch 32
run --timeset stream40 --batch 2
run --timeset stream40 --batch 2 --solver midpoint
run --timeset stream40 --batch 2 --solver rk4
run --timeset stream40 --batch 16 --solver rk4
run --timeset config1 --batch 8
cat $out
