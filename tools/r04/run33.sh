cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_conv_random.py -x -q -k "winograd" 2>&1 | tail -5 > gpurun_out/r04_z6_wino_tests.log
timeout 600 python tools/r04/winobench.py 5 2>&1 | grep '^{"layer' > gpurun_out/r04_z6_winobench.jsonl
export SF_LIB_PATH=build_r02/stamp/libsfnative.so
( timeout 300 python tools/r04/stamps_wino.py 64 64 32 200 200
  timeout 300 python tools/r04/stamps_wino.py 64 64 32 200 200 1 1
  timeout 300 python tools/r04/stamps_wino.py 128 128 32 200 200 ) > gpurun_out/r04_z6_stamps_wino.txt 2>&1
