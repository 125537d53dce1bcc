cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
SF_FLOW_TIMEOUT=65536 timeout 900 python -m pytest tests/test_gpu_persistent.py -m gpu -x -q 2>&1 | tail -3 > gpurun_out/r05_zb_tests.log
for v in head new head new; do
  lib=build_r02/sp_head/libsfnative.so
  [ $v = new ] && lib=streamingflow_amd/libsfnative.so
  SF_LIB_PATH=$lib SF_PERSIST=1 SF_FLOW_TIMEOUT=65536 timeout 600 python tools/chainbench.py euler 10 30 2>/dev/null | tail -1 | tr '\n' ' ' ; echo " flow $v"
done > gpurun_out/r05_zb_chain.txt 2>&1
