cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests/test_gpu_ops.py tests/test_gpu_forward.py tests/test_gpu_conv_random.py tests/test_gpu_persistent.py tests/test_gpu_end_to_end.py -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r05_w_tests.log
rm -f gpurun_out/r05_w_ab_group.jsonl
for k in 0 1 0 1; do
SF_WINO_GROUP=$k timeout 600 python bench.py --headline-only --no-roofline --steps 6 --warmup 2 2>/dev/null | tail -1 >> gpurun_out/r05_w_ab_group.jsonl
done
