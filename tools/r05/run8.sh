cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
for v in 1 0; do
  SF_WINO44=$v timeout 600 python tools/r04/winobench.py 5 2>gpurun_out/r05_h_err_$v.txt | grep -v '^{"winobench' > gpurun_out/r05_h_winobench_w44_$v.jsonl
done
for v in abl1 abl4 abl16 abl32; do
  for only in "DeepLab" "decoder / encoder 64->64"; do
    SF_LIB_PATH=build_r02/w5_$v/libsfnative.so WINOBENCH_ONLY="$only" timeout 300 python tools/r04/winobench.py 5 2>/dev/null | grep -v '^{"winobench' | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$v', d['layer'][:40], round(d['winograd_ms'], 3))"
  done
done > gpurun_out/r05_i_ablations44.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_conv_random.py -x -q -k "winograd" 2>&1 | tail -8 > gpurun_out/r05_h_wino_tests.log
