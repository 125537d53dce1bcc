cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r05_n_gputests.log
timeout 900 python bench.py --steps 10 --warmup 3 --headline-only > gpurun_out/r05_n_bench.json 2> gpurun_out/r05_n_bench.err
