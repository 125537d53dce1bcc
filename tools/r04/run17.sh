cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_conv_random.py -x -q -k "winograd" 2>&1 | tail -3 > gpurun_out/r04_p_wino_tests.log
timeout 600 python tools/r04/winobench.py 5 2>&1 | grep '^{"layer' > gpurun_out/r04_p_winobench.jsonl
bash tools/r04/pmc_wino.sh > gpurun_out/r04_p_pmc_wino.log 2>&1
cp gpurun_out/r04_pmc_wino.txt gpurun_out/r04_p_pmc_wino.txt
