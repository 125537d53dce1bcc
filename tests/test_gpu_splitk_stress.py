"""GPU: the fence-free (sc1 write-through) split-K hand-off against its fenced reference form and against no split at all
(ADVICE r2).  Each variant runs tools/splitk_stress.py in its own process (the switches are read once per process): 60
iterations x 5 layers reuse the same slab / ticket addresses with fresh inputs, cache sweeps and a concurrent stream."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(env_extra, dump):
    env = dict(os.environ)
    env.update(env_extra)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "splitk_stress.py"), "60", dump], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("CHECKSUMS")][0]
    return line.split()[1:]


def test_fence_free_handoff_equals_fenced_and_unsplit(tmp_path):
    a = _run({}, str(tmp_path / "a.pt"))
    b = _run({"SF_HANDOFF_FENCED": "1"}, str(tmp_path / "b.pt"))
    assert a == b, "fence-free and fenced split-K hand-offs differ bitwise"
    a2 = _run({}, str(tmp_path / "a2.pt"))
    assert a == a2, "the hand-off is not reproducible run to run"
    _run({"SF_SPLIT": "0"}, str(tmp_path / "c.pt"))
    ta, tc = torch.load(str(tmp_path / "a.pt")), torch.load(str(tmp_path / "c.pt"))
    for x, y in zip(ta, tc):       # another summation order: not bitwise, but the same numbers
        assert float((x - y).abs().max()) <= 2e-5
