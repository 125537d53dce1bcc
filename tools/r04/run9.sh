cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_conv_random.py -x -q -k "winograd" 2>&1 | tail -3 > gpurun_out/r04_i_wino_tests.log
for O in 1 0; do SF_WINO_OPTS=$O timeout 600 python tools/r04/winobench.py 5 2>&1 | grep '^{"layer' | sed "s/^/OPTS=$O /" ; done > gpurun_out/r04_i_winobench_opts.txt
