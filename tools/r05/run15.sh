cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
export SF_COMMIT=$(cat .sf_commit 2>/dev/null || echo unknown)
bash tools/r05/final_profile.sh > gpurun_out/r05_z_final_profile.log 2>&1
cd $GRAFT_REPO_ROOT
timeout 1200 python bench.py --steps 20 --warmup 3 > gpurun_out/r05_z_bench.json 2> gpurun_out/r05_z_bench.err
