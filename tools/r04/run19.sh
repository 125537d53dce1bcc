cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_conv_random.py -x -q -k "winograd" 2>&1 | tail -15 > gpurun_out/r04_r_wino_tests.log
WINOBENCH_ONLY=ASPP timeout 600 python tools/r04/winobench.py 5 2>&1 | grep '^{"layer' > gpurun_out/r04_r_winobench_dil.jsonl
timeout 900 python bench.py --steps 3 --warmup 1 > gpurun_out/r04_r_bench.json 2> gpurun_out/r04_r_bench.err
