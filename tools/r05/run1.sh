cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_cost tools/r05/mfma_cost.hip 2>/dev/null
timeout 300 /tmp/mfma_cost 2000 > gpurun_out/r05_a_mfma_cost.txt 2>&1
timeout 600 python tools/r04/winobench.py 5 > gpurun_out/r05_a_winobench_baseline.jsonl 2>&1
