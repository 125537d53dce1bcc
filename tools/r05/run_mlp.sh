cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q 2>&1 | tail -12 > gpurun_out/r05_n_mlp_tests.log
SF_MLP_FUSED=0 timeout 600 python bench.py --headline-only --steps 5 --warmup 2 > gpurun_out/r05_n_bench_two_launches.json 2> gpurun_out/r05_n_bench_two_launches.err
SF_MLP_FUSED=1 timeout 600 python bench.py --headline-only --steps 5 --warmup 2 > gpurun_out/r05_n_bench_fused.json 2> gpurun_out/r05_n_bench_fused.err
