cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
SF_WINO_TILE=2 timeout 600 python tools/r04/winobench.py 5 2>&1 | grep '^{"layer' | sed "s/^/TILE=2 /" > gpurun_out/r04_o_winobench_tile2.txt
SF_WINO_TILE=128 timeout 600 python tools/r04/winobench.py 5 2>&1 | grep '^{"layer' | head -5 | sed "s/^/TILE=128 /" >> gpurun_out/r04_o_winobench_tile2.txt
bash tools/r04/pmc_wino.sh > gpurun_out/r04_o_pmc_wino.log 2>&1
