cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_conv_random.py -x -q -k "winograd" 2>&1 | tail -4 > gpurun_out/r05_k_wino_tests.log
timeout 600 python tools/r04/winobench.py 5 2>/dev/null | grep -v '^{"winobench' > gpurun_out/r05_k_winobench.jsonl
timeout 1500 python -m pytest tests/test_gpu_forward.py tests/test_gpu_configs.py tests/test_gpu_ops.py tests/test_gpu_bf16x3.py -x -q 2>&1 | tail -4 > gpurun_out/r05_k_fwd_tests.log
SF_WINO_LIST=1 timeout 600 python3 tools/r05/wino_layers.py 32 2> gpurun_out/r05_k_wino_list_raw.txt >/dev/null
python3 - << 'PY'
lines = open("gpurun_out/r05_k_wino_list_raw.txt").read().split("\n")
i = max(k for k, l in enumerate(lines) if l.startswith("[sf-wino-begin]"))
open("gpurun_out/r05_k_wino_list_second_forward.txt", "w").write("\n".join(lines[i + 1:]))
PY
python3 tools/r05/wino_layers.py --summarise gpurun_out/r05_k_wino_list_second_forward.txt > gpurun_out/r05_k_wino_layers.txt
timeout 600 python3 tools/r05/sparse_fragment_density.py > gpurun_out/r05_k_sparse_fragment_density.jsonl 2> gpurun_out/r05_k_sparse_err.txt
timeout 600 python3 tools/sparsebench.py --cpu > gpurun_out/r05_k_sparsebench.json 2>> gpurun_out/r05_k_sparse_err.txt
