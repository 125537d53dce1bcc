#!/bin/bash
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $R/gpurun_out/pmc_clk -- python3 $R/tools/modbench.py --bigconvs > /dev/null 2>$R/gpurun_out/pmc_clk.err
