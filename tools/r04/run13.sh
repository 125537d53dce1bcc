cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
for T in 0 2; do SF_WINO_TILE=$T timeout 900 python -m pytest tests/test_gpu_conv_random.py -x -q -k "winograd" 2>&1 | tail -3 | sed "s/^/TILE=$T /"; done > gpurun_out/r04_l_wino_tests.log
for T in 0 2; do SF_WINO_TILE=$T timeout 600 python tools/r04/winobench.py 5 2>&1 | grep '^{"layer' | sed "s/^/TILE=$T /" ; done > gpurun_out/r04_l_winobench_tiles.txt
