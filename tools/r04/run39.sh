cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
export SF_COMMIT=${SF_COMMIT:-unknown}
rm -rf gpurun_out/pmcb_fetch gpurun_out/pmcb_write gpurun_out/final_trace
timeout 1800 bash tools/r04/final_profile.sh > gpurun_out/r04_z_final_profile.log 2>&1
timeout 1200 python bench.py > gpurun_out/r04_z_bench.json 2> gpurun_out/r04_z_bench.err
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -8 > gpurun_out/r04_z_gpu_tests.log
