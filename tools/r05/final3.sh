#!/bin/bash
# round-5 closing run: fabric traffic of the dominant kernel (PMC, separate passes), GPU tests, the driver's bench line, rocprofv3 kernel
# stats of the headline workload
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export SF_COMMIT=${SF_COMMIT:-unknown}
mkdir -p $R/gpurun_out
bash $R/tools/pmc_bench.sh > /dev/null 2>&1
python3 $R/tools/pmc_to_json.py "conv_wino5_kernel<0, false, " > $R/gpurun_out/pmc_dominant.log 2>&1
cp $R/profiles/pmc_dominant.json $R/gpurun_out/pmc_dominant.json
(cd $R && timeout 3000 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r05_zz_gputests.log)
(cd $R && timeout 1200 python3 bench.py > gpurun_out/r05_zz_bench.json 2> gpurun_out/r05_zz_bench.err)
rm -rf $R/gpurun_out/final_trace2
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final_trace2 -- python3 $R/bench.py --steps 5 --warmup 2 --headline-only > $R/gpurun_out/r05_zz_bench_headline_under_rocprof.json 2> $R/gpurun_out/final_trace2.err
cp $(ls $R/gpurun_out/final_trace2/*/*kernel_stats.csv | tail -1) $R/gpurun_out/r05_zz_kernel_stats_bench.csv
