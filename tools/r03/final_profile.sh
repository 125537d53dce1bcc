#!/bin/bash
# round-3 artefacts: fabric traffic of the dominant kernel and of the pipelined single-latent step (PMC, separate passes),
# rocprofv3 kernel stats of the bench workload, step traces
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export SF_COMMIT=${SF_COMMIT:-unknown}
# 1. dominant kernel of the forward
bash $R/tools/pmc_bench.sh > /dev/null 2>&1
python3 $R/tools/pmc_to_json.py > $R/gpurun_out/pmc_dominant.log 2>&1
cp $R/profiles/pmc_dominant.json $R/gpurun_out/pmc_dominant.json
# 2. GRU-ODE step: stand-alone calls at batch 8 / 200x200 as before, the single latent inside a rollout (chains of 10 and 30 steps)
N=20
for cfg in "8 50 50" "1 200 200"; do
  tag=$(echo $cfg | tr ' ' '_')
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $R/gpurun_out/pmcs_${c}_$tag
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmcs_${c}_$tag -- python3 $R/tools/stepbench.py $cfg $N > /dev/null 2>$R/gpurun_out/pmcs_${c}_$tag.err
  done
done
for n in 10 30; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $R/gpurun_out/pmcc_${c}_$n
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmcc_${c}_$n -- python3 $R/tools/chainrun.py $n 5 > /dev/null 2>$R/gpurun_out/pmcc_${c}_$n.err
  done
done
python3 $R/tools/pmc_step_to_json.py $N
cp $R/profiles/pmc_ode_step.json $R/gpurun_out/pmc_ode_step.json
# 3. kernel stats of the headline workload and of the step benches
rm -rf $R/gpurun_out/final_trace
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final_trace -- python3 $R/bench.py --steps 5 --warmup 2 --headline-only > $R/gpurun_out/final_trace_bench.json 2> $R/gpurun_out/final_trace.err
cp $(ls $R/gpurun_out/final_trace/*/*kernel_stats.csv | tail -1) $R/gpurun_out/r03_z_kernel_stats_bench.csv
bash $R/tools/r03/trace_chain.sh chain 9 > /dev/null 2>&1
for cfg in "8 50 50" "1 200 200"; do
  tag=$(echo $cfg | tr ' ' '_')
  rm -rf $R/gpurun_out/st_$tag
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/st_$tag -- python3 $R/tools/stepbench.py $cfg 50 > /dev/null 2>&1
  cp $(ls $R/gpurun_out/st_$tag/*/*kernel_stats.csv | tail -1) $R/gpurun_out/r03_z_kernel_stats_stepbench_$tag.csv
  python3 $R/tools/step_trace.py $(ls $R/gpurun_out/st_$tag/*/*kernel_trace.csv | tail -1) 15 > $R/gpurun_out/r03_z_step_trace_$tag.txt
done
ls $R/gpurun_out | head -50
