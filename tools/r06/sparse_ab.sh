#!/bin/bash
# SparseEncoder: rows sorted by neighbour mask + per-tile tap masks (SF_SPARSE_SORT) against the stored order, same box; then the parity tests
set -uo pipefail
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for k in 1 0 1 0; do
  echo "== SF_SPARSE_SORT=$k"
  SF_SPARSE_SORT=$k timeout 600 python3 -c "
import sys, json; sys.path.insert(0, 'tools')
import sparsebench
r = sparsebench.run(reps=5, cpu=False)
print(json.dumps({k: v for k, v in r.items() if not isinstance(v, dict)}))
print(json.dumps(r.get('roofline', {})))
" 2>&1 | tail -4
done
timeout 1500 python -m pytest tests/test_sparse_encoder.py tests/test_gpu_random_next_rows.py tests/test_gpu_end_to_end.py -q -m gpu -x 2>&1 | tail -4
