cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
SF_PROF_DUMP=1 SF_PW_STREAM=1 timeout 600 python bench.py --headline-only --steps 2 --warmup 1 > /dev/null 2> gpurun_out/r05_p_prof_dump_pw_on.txt
