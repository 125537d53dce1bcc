cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r04_a_gputests.log
SF_BENCH_BACKEND=gloo SF_BENCH_ONE_DEVICE=1 timeout 900 python bench.py --gpus 2 --batch 8 --steps 5 --warmup 2 > gpurun_out/r04_a_bench_2rank_gloo.log 2>&1
echo rc=$? >> gpurun_out/r04_a_bench_2rank_gloo.log
