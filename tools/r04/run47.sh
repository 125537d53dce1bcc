cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
export SF_LIB_PATH=build_r02/w16/libsfnative.so
SF_WINO_TILE=16 timeout 900 python -m pytest tests/test_gpu_conv_random.py -x -q -k "winograd_conv and not reproducible" 2>&1 | tail -5 > gpurun_out/r04_w16_tests.log
( echo "== 16 waves, 64 cout x 64 tiles, one workgroup per CU (SF_WINO_TILE=16)"; SF_WINO_TILE=16 timeout 600 python tools/r04/winobench.py 5 2>&1 | grep '^{"layer' | cut -c1-230
  echo "== default"; timeout 600 python tools/r04/winobench.py 5 2>&1 | grep '^{"layer' | cut -c1-230 ) > gpurun_out/r04_w16_winobench.txt
