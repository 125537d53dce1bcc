cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
for b in 32 48 64 96; do
  timeout 900 python bench.py --steps 4 --warmup 1 --headline-only --no-roofline --batch $b > gpurun_out/r04_batch_$b.json 2>gpurun_out/r04_batch_$b.err
done
python - <<'PY'
import json
for b in (32, 48, 64, 96):
    try:
        d=json.loads(open(f'gpurun_out/r04_batch_{b}.json').read().strip().split('\n')[-1])
        print(b, round(d['value'],1), round(d['ms_per_step'],2))
    except Exception as e:
        print(b, 'failed', e, open(f'gpurun_out/r04_batch_{b}.err').read()[-400:])
PY
