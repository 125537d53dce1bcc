# timing-only ablations of conv_wino5_kernel (build_var/w5_abl<bits>: tools/build_variant.sh w5_abl<bits> -DSF_W5_ABL=<bits>)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
for v in base abl1 abl2 abl4 abl8 abl16 abl32 abl63; do
  lib=build_var/w5_$v/libsfnative.so
  [ $v = base ] && lib=streamingflow_amd/libsfnative.so
  for only in "DeepLab" "decoder / encoder 64->64"; do
    SF_LIB_PATH=$lib WINOBENCH_ONLY="$only" timeout 300 python tools/r04/winobench.py 5 2>/dev/null | grep -v '^{"winobench' | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$v', d['layer'][:40], round(d['winograd_ms'], 3))"
  done
done > gpurun_out/r05_f_ablations.txt 2>&1
