#!/bin/bash
# MFMA-pipe utilisation and clock of the GRU-ODE step's kernels (one latent / 8 latents): SQ_VALU_MFMA_BUSY_CYCLES against
# GRBM_GUI_ACTIVE, per kernel and grid (tools/pmc_clock_summary.py).  Counters only beside --kernel-trace.
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "1 50 50" "8 50 50"; do
  tag=$(echo $cfg | tr ' ' '_')
  rm -rf $R/gpurun_out/pmc_clk_$tag
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $R/gpurun_out/pmc_clk_$tag -- python3 $R/tools/stepbench.py $cfg 20 > /dev/null 2>$R/gpurun_out/pmc_clk_$tag.err
  echo "== stepbench $cfg"
  python3 $R/tools/pmc_clock_summary.py $R/gpurun_out/pmc_clk_$tag
done
