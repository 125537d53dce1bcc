cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
SF_WINO_WHY=1 timeout 900 python bench.py --steps 1 --warmup 1 > gpurun_out/r04_q_bench_why.json 2> gpurun_out/r04_q_why.err
sort gpurun_out/r04_q_why.err | uniq -c | sort -rn | head -40 > gpurun_out/r04_q_why.txt
