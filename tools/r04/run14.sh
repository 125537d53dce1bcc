cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_conv_random.py -x -q -k "winograd" 2>&1 | tail -3 > gpurun_out/r04_m_wino_tests.log
timeout 600 python tools/r04/winobench.py 5 2>&1 | grep '^{"layer' > gpurun_out/r04_m_winobench.jsonl
timeout 600 python bench.py --steps 5 --warmup 2 --headline-only 2>&1 | grep '^{' > gpurun_out/r04_m_bench_headline.json
SF_WINO=0 timeout 600 python bench.py --steps 5 --warmup 2 --headline-only 2>&1 | grep '^{' > gpurun_out/r04_m_bench_headline_nowino.json
timeout 300 python tools/r04/sparse_tile_density.py 2>&1 | grep '^{' > gpurun_out/r04_m_sparse_tile_density.jsonl
