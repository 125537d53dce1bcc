cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for b in 5 6 7; do
  for p in 65536 8192; do
    echo -n "SF_WINO_MIN_P=$p  "; SF_WINO_MIN_P=$p timeout 300 python tools/chainbench.py euler 4 12 50 50 $b 2>&1 | grep "per step"
  done
done
for p in 65536 8192; do
  echo -n "SF_WINO_MIN_P=$p  "; SF_WINO_MIN_P=$p timeout 300 python tools/chainbench.py euler 4 12 200 200 1 2>&1 | grep "per step"
  echo -n "SF_WINO_MIN_P=$p  "; SF_WINO_MIN_P=$p timeout 300 python tools/chainbench.py euler 4 12 100 100 1 2>&1 | grep "per step"
done
