#!/bin/bash
# where do the waves of the large conv tiles spend their time?  separate passes, kernel-trace only
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_INST_LEVEL_LDS"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmcd_$i -- python3 $R/tools/modbench.py --bigconvs > /dev/null 2>$R/gpurun_out/pmcd_$i.err
done
python3 - <<'PY'
import csv, glob, os, collections
R = os.environ["GRAFT_REPO_ROOT"]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sorted(glob.glob(os.path.join(R, "gpurun_out", "pmcd_*"))):
    fs = glob.glob(os.path.join(d, "*", "*counter_collection.csv"))
    if not fs: continue
    for r in csv.DictReader(open(fs[-1])):
        if "conv_glds_kernel<4, 2, 2, 4, 0" in r["Kernel_Name"] and r["Grid_Size"] == "1120256":
            agg["LDS-DMA 128x128 tiles, 128->128 3x3 n7 (and the two other 7-frame 128-cout layers of --bigconvs)"][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if "conv_glds_kernel<2, 2, 2, 4, 0" in r["Kernel_Name"] and r["Grid_Size"] == "1120256":
            agg["LDS-DMA 64x128 tiles, 64->64 3x3 n7"][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(os.path.join(R, "gpurun_out", "pmc_diag.txt"), "w") as f:
    for k, v in agg.items():
        f.write(k + "\n")
        for c, vals in sorted(v.items()):
            f.write(f"  {c:34s} {sum(vals)/len(vals):16.0f}  (n={len(vals)})\n")
print(open(os.path.join(R, "gpurun_out", "pmc_diag.txt")).read())
PY
