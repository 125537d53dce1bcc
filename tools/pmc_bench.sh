#!/bin/bash
# HBM traffic of the dominant kernel of bench.py: FETCH_SIZE and WRITE_SIZE in separate passes
# (TCC slots), only --kernel-trace beside --pmc.  Post-processed by tools/pmc_to_json.py.
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
prog="python3 $R/bench.py --steps 2 --warmup 1 --headline-only --no-roofline"      # only the batch-32 forwards: per-launch averages over the launches bench.py prices
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmcb_fetch -- $prog > /dev/null 2>$R/gpurun_out/pmcb_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmcb_write -- $prog > /dev/null 2>$R/gpurun_out/pmcb_write.err
ls $R/gpurun_out/pmcb_fetch/*/ | head -3
