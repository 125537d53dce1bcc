#!/bin/bash
# round-end artefacts: HBM traffic of the dominant kernel (PMC) + rocprofv3 kernel stats of the bench workload
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
bash $R/tools/pmc_bench.sh > /dev/null 2>&1
python3 $R/tools/pmc_to_json.py > $R/gpurun_out/pmc_dominant.log 2>&1
cp $R/profiles/pmc_dominant.json $R/gpurun_out/pmc_dominant.json
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final_trace -- python3 $R/bench.py --steps 5 --warmup 2 --headline-only > $R/gpurun_out/final_trace_bench.json 2> $R/gpurun_out/final_trace.err
