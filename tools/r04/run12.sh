cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
SF_FLOW_TIMEOUT=16384 timeout 600 python -m pytest tests/test_gpu_persistent.py -x -q 2>&1 | tail -3 > gpurun_out/r04_k_persist_tests.log
for P in 0 1; do SF_FLOW_TIMEOUT=16384 SF_PERSIST=$P timeout 300 python tools/chainbench.py euler 10 30 2>&1 | grep chain | sed "s/^/SF_PERSIST=$P: /"; done > gpurun_out/r04_k_chain.log
for P in 0 1; do SF_FLOW_TIMEOUT=16384 SF_PERSIST=$P timeout 300 python tools/chainbench.py rk4 4 10 2>&1 | grep chain | sed "s/^/SF_PERSIST=$P: /"; done >> gpurun_out/r04_k_chain.log
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 > gpurun_out/r04_k_gputests.log
