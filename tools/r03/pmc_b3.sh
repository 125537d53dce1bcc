#!/bin/bash
# where do the waves of the bf16x3 128x128 tiles spend their time?  separate PMC passes, kernel-trace only
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mode=${1:-bf16x3}
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/pmcb3_$i
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmcb3_$i -- python3 $R/tools/r03/b3_quick.py 8 $mode > /dev/null 2>$R/gpurun_out/pmcb3_$i.err
done
python3 - $mode <<'PY'
import csv, glob, os, collections, sys
R = os.environ["GRAFT_REPO_ROOT"]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sorted(glob.glob(os.path.join(R, "gpurun_out", "pmcb3_*"))):
    fs = glob.glob(os.path.join(d, "*", "*counter_collection.csv"))
    if not fs: continue
    for r in csv.DictReader(open(fs[-1])):
        for key in ("conv_glds_kernel<4, 2, 2, 4, 0", "conv_glds_kernel<2, 2, 2, 4, 0", "conv_glds_kernel<4, 2, 1, 4, 2", "conv_glds_kernel<2, 2, 2, 2, 0, 2, false"):
            if key in r["Kernel_Name"]:
                agg[key + " (all launches of the batch-8 forward)"][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = os.path.join(R, "gpurun_out", f"r03_pmc_diag_{sys.argv[1]}.txt")
with open(out, "w") as f:
    for k, v in agg.items():
        f.write(k + "\n")
        for c, vals in sorted(v.items()):
            f.write(f"  {c:34s} {sum(vals)/len(vals):16.0f}  (n={len(vals)})\n")
print(open(out).read())
PY
