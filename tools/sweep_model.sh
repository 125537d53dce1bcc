#!/bin/bash
out=gpurun_out/sweep_model.txt
: > $out
for d in 0 1; do
  echo "== SF_SPLIT_MODEL=$d" >> $out
  SF_SPLIT_MODEL=$d python tools/modbench.py --quick 2>/dev/null | grep -E "rollout|dual|infer" >> $out
  SF_SPLIT_MODEL=$d python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('bench value',d['value'],'single_ms',d['single_sample_forward_ms'],'step_only', {k:(v['us_per_step'],v['tflops']) for k,v in d['ode_step_only'].items() if isinstance(v,dict) and 'us_per_step' in v})" >> $out
done
SF_SPLIT_MODEL=1 python -m pytest tests/test_gpu_forward.py tests/test_gpu_ops.py -x -q -m gpu 2>&1 | tail -2 >> $out
