cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
rm -rf $R/gpurun_out/trace_b1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/trace_b1 -- python3 $R/bench.py --steps 20 --warmup 3 --batch 1 --headline-only --no-roofline > $R/gpurun_out/r05_b1_bench.json 2> $R/gpurun_out/trace_b1.err
cp $(ls $R/gpurun_out/trace_b1/*/*kernel_stats.csv | tail -1) $R/gpurun_out/r05_b1_kernel_stats.csv
