cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python bench.py > gpurun_out/r04_z_bench.json 2> gpurun_out/r04_z_bench.err
