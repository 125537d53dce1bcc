cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
SF_FLOW_TIMEOUT=16384 timeout 600 python -m pytest tests/test_gpu_persistent.py -x -q 2>&1 | tail -3 > gpurun_out/r04_c_persist_tests.log
for P in 0 1; do SF_FLOW_TIMEOUT=16384 SF_PERSIST=$P timeout 300 python tools/chainbench.py euler 10 30 2>&1 | grep chain; done > gpurun_out/r04_c_chain.log
SF_FLOW_TIMEOUT=16384 SF_PERSIST=1 SF_FLOW_SC1=1 timeout 300 python tools/chainbench.py euler 10 30 2>&1 | grep chain | sed 's/^/SF_FLOW_SC1=1 (no acquire, results may be stale): /' >> gpurun_out/r04_c_chain.log
SF_FLOW_TIMEOUT=16384 SF_PERSIST=1 SF_LIB_PATH=build_r02/stamp/libsfnative.so timeout 300 python tools/r04/flow_stamps.py 5 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_c_flow_stamps.txt
