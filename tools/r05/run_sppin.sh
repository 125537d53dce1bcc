cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_gpu_persistent.py tests/test_gpu_forward.py tests/test_gpu_ops.py -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r05_z9_tests.log
for v in nopin pin nopin pin; do
  lib=build_r02/sp_nopin/libsfnative.so
  [ $v = pin ] && lib=streamingflow_amd/libsfnative.so
  SF_LIB_PATH=$lib timeout 600 python tools/chainbench.py euler 10 30 2>/dev/null | tail -1 | tr '\n' ' ' ; echo " $v"
done > gpurun_out/r05_z9_chain.txt 2>&1
