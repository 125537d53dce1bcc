cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
for v in 1 0; do
  SF_WINO44=$v timeout 600 python tools/r04/winobench.py 5 2>gpurun_out/r05_h_err_$v.txt | grep -v '^{"winobench' > gpurun_out/r05_h_winobench_w44_$v.jsonl
done
timeout 900 python -m pytest tests/test_gpu_conv_random.py -x -q -k "winograd" 2>&1 | tail -8 > gpurun_out/r05_h_wino_tests.log
timeout 1500 python -m pytest tests/test_gpu_forward.py tests/test_gpu_configs.py tests/test_gpu_ops.py -x -q 2>&1 | tail -8 > gpurun_out/r05_h_fwd_tests.log
