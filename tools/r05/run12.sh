cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
SF_PROF_DUMP=1 timeout 600 python bench.py --steps 2 --warmup 1 --headline-only > /dev/null 2> gpurun_out/r05_m_prof_dump.txt
grep "sf-prof" gpurun_out/r05_m_prof_dump.txt | python3 -c "
import sys, collections
agg = collections.OrderedDict()
for l in sys.stdin:
    kv = dict(x.split('=') for x in l.split()[1:])
    k = (int(kv['key']), kv['gflop'], kv['mbytes'])
    a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += float(kv['us'])
tot = sum(a[1] for a in agg.values())
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f'key {k[0]:4d} gflop {float(k[1]):9.2f} MB {float(k[2]):9.1f}  x{a[0]:4d}  {a[1] / a[0]:9.1f} us each  {float(k[1]) / (a[1] / a[0]) * 1e3 / 157.3:5.3f} of mfma peak  {float(k[2]) / (a[1] / a[0]) / 8e3 * 1e0:5.3f} of 8 TB/s')
" > gpurun_out/r05_m_prof_by_launch.txt
