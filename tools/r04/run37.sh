cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -8 > gpurun_out/r04_zz_gpu_tests.log
timeout 1200 python bench.py > gpurun_out/r04_zz_bench.json 2> gpurun_out/r04_zz_bench.err
