cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -8 > gpurun_out/r04_z_gpu_tests.log
timeout 1200 python bench.py > gpurun_out/r04_z_bench.json 2> gpurun_out/r04_z_bench.err
export SF_COMMIT=e1339a0
rm -rf gpurun_out/pmcb_fetch gpurun_out/pmcb_write
timeout 1800 bash tools/r04/final_profile.sh > gpurun_out/r04_z_final_profile.log 2>&1
bash tools/r04/pmc_wino.sh > gpurun_out/r04_z_pmc_wino.log 2>&1
cp gpurun_out/r04_pmc_wino.txt gpurun_out/r04_z_pmc_wino_sq_counters.txt
timeout 600 python tools/r04/winobench.py 5 2>&1 | grep '^{"layer' > gpurun_out/r04_z_winobench.jsonl
