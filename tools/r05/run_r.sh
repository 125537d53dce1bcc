cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
for k in 0 1 0 1; do
SF_HEAD_PLANAR=$k timeout 600 python bench.py --headline-only --no-roofline --steps 6 --warmup 2 2>/dev/null | tail -1 >> gpurun_out/r05_r_ab_planar.jsonl
done
