cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_conv_random.py -x -q -k "winograd" 2>&1 | tail -5 > gpurun_out/r04_u_wino_tests.log
timeout 600 python tools/r04/winobench.py 5 2>&1 | grep '^{"layer' > gpurun_out/r04_u_winobench.jsonl
timeout 900 python bench.py --steps 3 --warmup 1 > gpurun_out/r04_u_bench.json 2> gpurun_out/r04_u_bench.err
