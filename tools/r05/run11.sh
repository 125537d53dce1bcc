cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
bash tools/r05/abl_weight_dma.sh > gpurun_out/r05_l_ablation_step_without_weight_dmas.txt 2>&1
cd $GRAFT_REPO_ROOT
timeout 900 python bench.py --steps 10 --warmup 3 > gpurun_out/r05_l_bench.json 2> gpurun_out/r05_l_bench.err
