cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests/test_gpu_ops.py tests/test_gpu_forward.py tests/test_gpu_persistent.py tests/test_gpu_conv_random.py -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r05_z1_tests.log
for k in 0 1 0 1; do
  SF_SP_MAGIC=$k timeout 600 python tools/chainbench.py euler 10 30 2>/dev/null | tail -3 | tr '\n' ' ' ; echo " SF_SP_MAGIC=$k"
done > gpurun_out/r05_z1_chain.txt 2>&1
