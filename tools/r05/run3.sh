cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
for cfg in "64 64 32 200 200" "128 128 32 200 200" "128 64 32 200 200 12"; do
  SF_LIB_PATH=build_r02/stamp/libsfnative.so timeout 300 python3 tools/r04/stamps_wino.py $cfg
done > gpurun_out/r05_c_stamps_wino5.txt 2>&1
bash tools/r05/pmc_wino.sh > gpurun_out/r05_c_pmc_wino5.log 2>&1
