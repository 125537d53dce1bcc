cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_persistent.py -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r05_z6_tests.log
for k in 0 1 0 1; do
  SF_PERSIST=$k SF_FLOW_TIMEOUT=65536 timeout 600 python tools/chainbench.py euler 10 30 2>/dev/null | tail -1 | tr '\n' ' ' ; echo " SF_PERSIST=$k"
done > gpurun_out/r05_z6_chain.txt 2>&1
