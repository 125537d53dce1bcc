cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/r05_x_batch_sweep.jsonl
for b in 16 24 32 40 48 64; do
timeout 900 python bench.py --headline-only --no-roofline --steps 4 --warmup 2 --batch $b 2>gpurun_out/r05_x_batch_$b.err | tail -1 >> gpurun_out/r05_x_batch_sweep.jsonl
done
