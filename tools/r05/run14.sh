cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_conv_random.py -x -q -k "winograd" 2>&1 | tail -4 > gpurun_out/r05_o_wino_tests.log
timeout 1500 python -m pytest tests/test_gpu_forward.py tests/test_gpu_configs.py tests/test_gpu_ops.py tests/test_beverse.py tests/test_unused_cells.py -x -q 2>&1 | tail -4 > gpurun_out/r05_o_fwd_tests.log
for f in 1 0; do
SF_WINO_FUSE_DEC=$f timeout 900 python bench.py --steps 10 --warmup 3 --headline-only > gpurun_out/r05_o_bench_fuse$f.json 2> gpurun_out/r05_o_bench.err
done
