cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_voxel.py tests/test_beverse.py tests/test_cabi.py tests/test_gpu_cabi_c.py -q -m gpu 2>&1 | tail -8 > gpurun_out/r05_q_tests.log
