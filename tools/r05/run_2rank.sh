cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
SF_BENCH_BACKEND=gloo SF_BENCH_ONE_DEVICE=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --batch 8 --steps 5 --warmup 2 --headline-only --no-roofline > gpurun_out/r05_ze_bench_2rank_gloo_one_gpu.json 2> gpurun_out/r05_ze_bench_2rank.err
