cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_conv_random.py -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r05_y_tests.log
for v in nomagic magic nomagic magic; do
  lib=build_r02/w5_nomagic/libsfnative.so
  [ $v = magic ] && lib=streamingflow_amd/libsfnative.so
  SF_LIB_PATH=$lib timeout 600 python tools/r04/winobench.py 5 2>/dev/null | grep -v '^{"winobench' | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$v', d['layer'][:44], round(d['winograd_ms'], 3))"
done > gpurun_out/r05_y_magic.txt 2>&1
for v in nomagic magic nomagic magic; do
  lib=build_r02/w5_nomagic/libsfnative.so
  [ $v = magic ] && lib=streamingflow_amd/libsfnative.so
  SF_LIB_PATH=$lib timeout 600 python bench.py --headline-only --no-roofline --steps 6 --warmup 2 2>/dev/null | tail -1 | python3 -c "
import sys, json
d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'])" >> gpurun_out/r05_y_magic.txt
done
