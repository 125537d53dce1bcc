cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
SF_LIB_PATH=build_r02/stamp/libsfnative.so timeout 600 python tools/r02/stamps.py 1 50 50 > gpurun_out/r05_z4_stamps_step.txt 2>&1
