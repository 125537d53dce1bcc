cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_conv_random.py -m gpu -x -q 2>&1 | tail -12 > gpurun_out/r05_p_pw_tests.log
SF_PW_STREAM=0 timeout 600 python bench.py --headline-only --steps 5 --warmup 2 > gpurun_out/r05_p_bench_pw_off.json 2> gpurun_out/r05_p_bench_pw_off.err
SF_PW_STREAM=1 timeout 600 python bench.py --headline-only --steps 5 --warmup 2 > gpurun_out/r05_p_bench_pw_on.json 2> gpurun_out/r05_p_bench_pw_on.err
