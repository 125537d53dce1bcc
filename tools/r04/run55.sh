cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_conv_random.py -x -q -k "winograd" 2>&1 | tail -5 > gpurun_out/r04_cat_wino_tests.log
timeout 900 python -m pytest tests/test_gpu_forward.py tests/test_gpu_configs.py tests/test_gpu_ops.py -x -q 2>&1 | tail -5 > gpurun_out/r04_cat_fwd_tests.log
for c in 1 0; do
  echo "== SF_WINO_CAT=$c"
  SF_WINO_CAT=$c timeout 600 python bench.py --steps 6 --warmup 2 --headline-only 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(round(d['value'],1), round(d['ms_per_step'],2), {k:(v['calls_per_forward'], round(v['ms_per_forward'],2)) for k,v in d['roofline']['per_kernel'].items() if 'wino' in k})"
  for b in 8 16; do SF_WINO_CAT=$c timeout 300 python tools/chainbench.py euler 4 12 50 50 $b 2>&1 | grep "per step"; done
done > gpurun_out/r04_cat_bench.txt 2>&1
