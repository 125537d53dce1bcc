#!/bin/bash
# Round-2 artefacts in one GPU call: PMC traffic (ODE step + dominant bench kernel), rocprofv3 kernel stats of the bench
# and of the step bench, the default bench line.  Outputs under gpurun_out/; the summaries are copied into profiles/ by hand.
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
bash $R/tools/pmc_step.sh > $R/gpurun_out/pmc_step.log 2>&1
bash $R/tools/final_profile.sh
for cfg in "1 50 50" "8 50 50" "1 200 200"; do
  tag=$(echo $cfg | tr ' ' '_')
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/stepstats_$tag -- python3 $R/tools/stepbench.py $cfg 50 > $R/gpurun_out/stepstats_$tag.log 2>$R/gpurun_out/stepstats_$tag.err
done
cd $R
python3 bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
tail -c 600 gpurun_out/bench_default.json
