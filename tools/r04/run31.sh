cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_conv_random.py -x -q -k "winograd" 2>&1 | tail -5 > gpurun_out/r04_z4_wino_tests.log
( echo "== rolling transform"; timeout 600 python tools/r04/winobench.py 5 2>&1 | grep '^{"layer' | cut -c1-230
  echo "== one-shot transform (-DSF_WINO_NOROLL)"; SF_LIB_PATH=build_r02/noroll/libsfnative.so timeout 600 python tools/r04/winobench.py 5 2>&1 | grep '^{"layer' | cut -c1-230 ) > gpurun_out/r04_z4_winobench_roll.txt
