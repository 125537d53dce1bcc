#!/bin/bash
# round-5 closing run after the fused ConvNeXt MLP: GPU tests, the driver's bench line, rocprofv3 kernel stats of the headline workload
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
(cd $R && timeout 3000 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r05_zz_gputests.log)
(cd $R && timeout 1200 python3 bench.py > gpurun_out/r05_zz_bench.json 2> gpurun_out/r05_zz_bench.err)
rm -rf $R/gpurun_out/final_trace2
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final_trace2 -- python3 $R/bench.py --steps 5 --warmup 2 --headline-only > $R/gpurun_out/r05_zz_bench_headline_under_rocprof.json 2> $R/gpurun_out/final_trace2.err
cp $(ls $R/gpurun_out/final_trace2/*/*kernel_stats.csv | tail -1) $R/gpurun_out/r05_zz_kernel_stats_bench.csv
