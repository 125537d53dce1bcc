cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_conv_random.py -x -q -k "winograd" 2>&1 | tail -8 > gpurun_out/r05_b_wino_tests.log
for v in 5 4; do
  SF_WINO_V=$v timeout 600 python tools/r04/winobench.py 5 2>/dev/null | grep -v '^{"winobench' > gpurun_out/r05_b_winobench_v$v.jsonl
done
timeout 1500 python -m pytest tests/test_gpu_forward.py tests/test_gpu_configs.py tests/test_gpu_ops.py -x -q 2>&1 | tail -8 > gpurun_out/r05_b_fwd_tests.log
