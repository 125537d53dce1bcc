#!/bin/bash
# PMC passes (one counter group per run, only --kernel-trace beside --pmc, as gpurun requires)
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
prog="python3 $R/tools/modbench.py --bigconvs"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU --output-format csv -d $R/gpurun_out/pmc_sq -- $prog > /dev/null 2>$R/gpurun_out/pmc_sq.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE --output-format csv -d $R/gpurun_out/pmc_sq2 -- $prog > /dev/null 2>$R/gpurun_out/pmc_sq2.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -- $prog > /dev/null 2>$R/gpurun_out/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -- $prog > /dev/null 2>$R/gpurun_out/pmc_write.err
ls -R $R/gpurun_out/pmc_sq | head
