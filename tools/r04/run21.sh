cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_conv_random.py -x -q -k "winograd" 2>&1 | tail -5 > gpurun_out/r04_t_wino_tests.log
( for st in 0 1 2 3 4; do echo "== stagger $st"; SF_WINO_STAGGER=$st timeout 600 python tools/r04/winobench.py 5 2>&1 | grep '^{"layer' | cut -c1-230; done ) > gpurun_out/r04_t_winobench_stagger.txt
