cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
(hipcc --offload-arch=gfx950 -O2 tools/r04/kernarg_probe.hip -o /tmp/kp && timeout 120 /tmp/kp) > gpurun_out/r04_b_kernarg_probe.txt 2>&1
SF_FLOW_TIMEOUT=16384 timeout 900 python -m pytest tests/test_gpu_persistent.py -x -q > gpurun_out/r04_b_persist_tests.log 2>&1
echo "rc=$?" >> gpurun_out/r04_b_persist_tests.log
for P in 0 1; do SF_FLOW_TIMEOUT=16384 SF_PERSIST=$P timeout 300 python tools/chainbench.py euler 10 30; done > gpurun_out/r04_b_chain.log 2>&1
for M in 1 2 4 9; do SF_FLOW_TIMEOUT=16384 SF_PERSIST=1 SF_SEG_MAXPH=$M timeout 300 python tools/chainbench.py euler 10 30; done >> gpurun_out/r04_b_chain.log 2>&1
