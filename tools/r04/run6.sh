cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_conv_random.py -x -q -k "winograd or dma_conv" 2>&1 | tail -15 > gpurun_out/r04_f_wino_tests.log
timeout 600 python tools/r04/winobench.py 5 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_f_winobench.log
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -25 > gpurun_out/r04_f_gputests.log
