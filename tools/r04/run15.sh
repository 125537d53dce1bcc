cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 > gpurun_out/r04_n_gputests.log
timeout 600 python tools/r04/winobench.py 5 2>&1 | grep '^{"layer' > gpurun_out/r04_n_winobench.jsonl
timeout 900 python bench.py > gpurun_out/r04_n_bench_full.log 2>&1
