#!/bin/bash
# round-5 artefacts: fabric traffic of the dominant kernel and of the single-latent step (launch per layer and persistent flow kernel;
# PMC, separate passes, --kernel-trace only beside --pmc), rocprofv3 kernel stats of the bench workload, step traces in both forms
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export SF_COMMIT=${SF_COMMIT:-unknown}
export SF_FLOW_TIMEOUT=65536
# 1. dominant kernel of the forward
bash $R/tools/pmc_bench.sh > /dev/null 2>&1
python3 $R/tools/pmc_to_json.py "conv_wino5_kernel<0, false, " > $R/gpurun_out/pmc_dominant.log 2>&1      # bench.py's dominant kernel (most time in the batched forward)
cp $R/profiles/pmc_dominant.json $R/gpurun_out/pmc_dominant.json
# 2. the single latent inside a rollout: chains of 10 and 30 steps, both forms
for n in 10 30; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $R/gpurun_out/pmcc_${c}_$n $R/gpurun_out/pmcf_${c}_$n
    SF_PERSIST=0 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmcc_${c}_$n -- python3 $R/tools/chainrun.py $n 5 > /dev/null 2>$R/gpurun_out/pmcc_${c}_$n.err
    SF_PERSIST=1 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmcf_${c}_$n -- python3 $R/tools/chainrun.py $n 5 > /dev/null 2>$R/gpurun_out/pmcf_${c}_$n.err
  done
done
python3 $R/tools/pmc_step_to_json.py 20 > $R/gpurun_out/pmc_ode_step.log 2>&1
cp $R/profiles/pmc_ode_step.json $R/gpurun_out/pmc_ode_step.json
# 3. kernel stats of the headline workload
rm -rf $R/gpurun_out/final_trace
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final_trace -- python3 $R/bench.py --steps 5 --warmup 2 --headline-only > $R/gpurun_out/r05_z_bench_headline_under_rocprof.json 2> $R/gpurun_out/final_trace.err
cp $(ls $R/gpurun_out/final_trace/*/*kernel_stats.csv | tail -1) $R/gpurun_out/r05_z_kernel_stats_bench.csv
# 4. per-launch timeline of the steady-state step, both forms
for P in 0 1; do
  rm -rf $R/gpurun_out/trace_chain_$P
  SF_PERSIST=$P rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/trace_chain_$P -- python3 $R/tools/chainbench.py euler 4 8 > $R/gpurun_out/trace_chain_$P.log 2>&1
  cp $(ls $R/gpurun_out/trace_chain_$P/*/*kernel_stats.csv | tail -1) $R/gpurun_out/r05_z_kernel_stats_chain_SF_PERSIST$P.csv
  python3 $R/tools/step_trace.py $(ls $R/gpurun_out/trace_chain_$P/*/*kernel_trace.csv | tail -1) 9 tail > $R/gpurun_out/r05_z_step_trace_in_rollout_SF_PERSIST$P.txt 2>&1
done
ls $R/gpurun_out | grep r04_z
