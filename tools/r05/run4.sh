cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
SF_WINO_LIST=1 timeout 600 python3 tools/r05/wino_layers.py 32 2> gpurun_out/r05_d_wino_list_raw.txt >/dev/null
python3 - << 'PY'
lines = open("gpurun_out/r05_d_wino_list_raw.txt").read().split("\n")
i = max(k for k, l in enumerate(lines) if l.startswith("[sf-wino-begin]"))
open("gpurun_out/r05_d_wino_list_second_forward.txt", "w").write("\n".join(lines[i + 1:]))
PY
python3 tools/r05/wino_layers.py --summarise gpurun_out/r05_d_wino_list_second_forward.txt > gpurun_out/r05_d_wino_layers.txt
