cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
for v in base tinter prio both; do
  lib=build_r02/w5_$v/libsfnative.so
  [ $v = base ] && lib=streamingflow_amd/libsfnative.so
  SF_LIB_PATH=$lib timeout 600 python tools/r04/winobench.py 5 2>/dev/null | grep -v '^{"winobench' > gpurun_out/r05_e_winobench_$v.jsonl
done
