cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_conv_random.py -x -q -k "winograd" 2>&1 | tail -5 > gpurun_out/r04_z7_wino_tests.log
( echo "== persistent (two workgroups per CU loop over their tiles)"; timeout 600 python tools/r04/winobench.py 5 2>&1 | grep '^{"layer' | cut -c1-230
  echo "== one workgroup per tile (SF_WINO_PERSIST=0)"; SF_WINO_PERSIST=0 timeout 600 python tools/r04/winobench.py 5 2>&1 | grep '^{"layer' | cut -c1-230 ) > gpurun_out/r04_z7_winobench_persist.txt
