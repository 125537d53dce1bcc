#!/bin/bash
# round-end artefacts of one lease: the full bench line, the rocprofv3 kernel statistics of the same command (headline only), the HBM-traffic
# passes of the dominant kernel and of the ODE step (PMC; separate passes, only --kernel-trace beside --pmc), the step's launch timeline.
set -uo pipefail
: "${GRAFT_REPO_ROOT:?run through gpurun}"
R=$GRAFT_REPO_ROOT
tag=${1:-r06_z}
export SF_COMMIT=${2:-unknown}
cd $R; export TMPDIR=/tmp
timeout 1500 python3 bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
cd /tmp
rm -rf $R/gpurun_out/${tag}_trace
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_trace -- python3 $R/bench.py --steps 5 --warmup 2 --headline-only > $R/gpurun_out/${tag}_bench_headline_under_rocprof.json 2> $R/gpurun_out/${tag}_trace.err
cp $(ls $R/gpurun_out/${tag}_trace/*/*kernel_stats.csv | tail -1) $R/gpurun_out/${tag}_kernel_stats_bench.csv
bash $R/tools/pmc_bench.sh > /dev/null 2>&1
python3 $R/tools/pmc_to_json.py "conv_wino5_kernel<0, false, " > $R/gpurun_out/${tag}_pmc_dominant.log 2>&1
cp $R/profiles/pmc_dominant.json $R/gpurun_out/pmc_dominant.json
# the single latent inside a rollout: chains of 10 and 30 steps (tools/pmc_step_to_json.py takes the difference), launch path
for n in 10 30; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $R/gpurun_out/pmcc_${c}_$n
    SF_PERSIST=0 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmcc_${c}_$n -- python3 $R/tools/chainrun.py $n 5 > /dev/null 2>$R/gpurun_out/pmcc_${c}_$n.err
  done
done
export SF_COMMIT=${SF_COMMIT:-unknown}
bash $R/tools/pmc_step.sh > $R/gpurun_out/${tag}_pmc_step.log 2>&1
cp $R/profiles/pmc_ode_step.json $R/gpurun_out/pmc_ode_step.json
rm -rf $R/gpurun_out/${tag}_chain
SF_PERSIST=0 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_chain -- python3 $R/tools/chainbench.py euler 4 8 > $R/gpurun_out/${tag}_chain.log 2>&1
python3 $R/tools/step_trace.py $(ls $R/gpurun_out/${tag}_chain/*/*kernel_trace.csv | tail -1) 9 tail > $R/gpurun_out/${tag}_step_trace_in_rollout.txt
tail -1 $R/gpurun_out/${tag}_bench.json | cut -c 1-600
