cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_forward.py tests/test_gpu_configs.py tests/test_gpu_ops.py -x -q 2>&1 | tail -5 > gpurun_out/r04_sc_fwd_tests.log
SF_WINO_WHY=1 timeout 600 python bench.py --steps 6 --warmup 2 --headline-only > gpurun_out/r04_sc_bench.json 2> gpurun_out/r04_sc_why.err
sort gpurun_out/r04_sc_why.err | uniq -c | sort -rn | head -12 > gpurun_out/r04_sc_why.txt
