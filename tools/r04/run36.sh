cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
for p in 131072 60000 30000; do
  SF_WINO_MIN_P=$p timeout 600 python bench.py --steps 6 --warmup 2 --headline-only > gpurun_out/r04_minp_$p.json 2>/dev/null
done
python - <<'PY'
import json
for p in (131072, 60000, 30000):
    d=json.loads(open(f'gpurun_out/r04_minp_{p}.json').read().strip().split('\n')[-1])
    pk=d['roofline']['per_kernel']
    print(p, round(d['value'],1), round(d['ms_per_step'],2), {k:(v['calls_per_forward'], round(v['ms_per_forward'],2)) for k,v in pk.items() if 'wino' in k or '64x64' in k})
PY
