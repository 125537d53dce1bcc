cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_forward.py -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r05_v_tests.log
rm -f gpurun_out/r05_v_ab_dw.jsonl
for k in 0 1 0 1; do
SF_DWCONV_PK=$k timeout 600 python bench.py --headline-only --no-roofline --steps 6 --warmup 2 2>/dev/null | tail -1 >> gpurun_out/r05_v_ab_dw.jsonl
done
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/trace_v -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --headline-only --no-roofline > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/trace_v.err
grep dwconv $(ls $GRAFT_REPO_ROOT/gpurun_out/trace_v/*/*kernel_stats.csv | tail -1) > $GRAFT_REPO_ROOT/gpurun_out/r05_v_dwconv_stats.txt
