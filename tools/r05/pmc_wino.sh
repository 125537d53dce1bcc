#!/bin/bash
# where do the waves of the Winograd kernels spend their time?  SQ counters in separate passes (--kernel-trace only beside --pmc)
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "128 0 128 64 200 200 6" "64 0 64 64 200 200 6"; do
  tag=$(echo $cfg | cut -d' ' -f1,3 | tr ' ' '_')
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
             "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
             "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_SALU" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS" \
             "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM"; do
    i=$((i+1))
    rm -rf $R/gpurun_out/pmcw_${tag}_$i
    rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmcw_${tag}_$i -- python3 $R/tools/r04/winolayer.py $cfg > /dev/null 2>$R/gpurun_out/pmcw_${tag}_$i.err
  done
done
python3 - <<'PY'
import csv, glob, os, collections
R = os.environ["GRAFT_REPO_ROOT"]
out = open(os.path.join(R, "gpurun_out", "r05_pmc_wino.txt"), "w")
for tag in ("128_128", "64_64"):
    agg = collections.defaultdict(list)
    dur = []
    for d in sorted(glob.glob(os.path.join(R, "gpurun_out", f"pmcw_{tag}_*"))):
        fs = glob.glob(os.path.join(d, "*", "*counter_collection.csv"))
        if not fs: continue
        for r in csv.DictReader(open(fs[-1])):
            if "conv_wino" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
                name = r["Kernel_Name"]
        kt = glob.glob(os.path.join(d, "*", "*kernel_trace.csv"))
        if kt:
            for r in csv.DictReader(open(kt[-1])):
                if "conv_wino" in r["Kernel_Name"]: dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    if not agg: continue
    out.write(f"{tag}: {name[:90]}  avg {sum(dur)/max(1,len(dur))/1e3:.1f} us per launch under PMC\n")
    for c, vals in sorted(agg.items()):
        out.write(f"  {c:34s} {sum(vals)/len(vals):18.0f}  (n={len(vals)})\n")
out.close()
print(open(os.path.join(R, "gpurun_out", "r05_pmc_wino.txt")).read())
PY
