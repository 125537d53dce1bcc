cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r05_g_gputests.log
timeout 900 python bench.py --steps 10 --warmup 3 > gpurun_out/r05_g_bench.json 2> gpurun_out/r05_g_bench.err
