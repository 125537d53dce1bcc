"""GPU: persistent segment launches of the single-latent rollout (conv_sp.hip: sp_segment_kernel; north star: the stepping loop
fused into one LDS-tiled kernel) against the launch-per-layer path.  Same tiles, same K slices, same summation orders: the two
must agree BIT FOR BIT — any stale read across the in-kernel phase hand-off (sc1 stores -> counter -> acquire) shows up here."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import sys, torch
sys.path.insert(0, %r); sys.path.insert(0, %r)
from util import build_pair, cases, hashfill
from streamingflow_amd import schedule as S
import streamingflow_amd as sfa
solver, ts, mode, out = sys.argv[1:5]
sfa.set_math_mode(mode)
C, h, w = 64, 50, 50
cts, lts, tts, dt = cases.timeset(ts)
net, _ = build_pair(C, solver, True, True, dt)
ode = net.gru_ode
times, _ = S.merge_observations(cts[0].tolist(), lts[0].tolist())
sc = S.build_schedule(times, tts[0].tolist(), dt, True, solver)
res = []
for k in range(3):
    hx = (hashfill.normal(f"pshx{k}", (len(times), h, w, C), 61) * 0.5).cuda()
    eps = hashfill.normal(f"pseps{k}", (sc.n_draws, h, w, C), 62).cuda()
    ode.use_graph = (k == 2)
    a, fa = ode.rollout_nhwc(hx, sc, eps)
    res += [a.cpu().clone(), fa.cpu().clone()]
    torch.empty(64 << 20, device="cuda").fill_(1.0)      # sweep the caches between rollouts
torch.save(res, out)
""" % (ROOT, os.path.join(ROOT, "tests"))


def _run(solver, ts, mode, persist, out):
    env = dict(os.environ)
    env["SF_PERSIST"] = str(persist)
    r = subprocess.run([sys.executable, "-c", SCRIPT, solver, ts, mode, out], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    return torch.load(out)


@pytest.mark.parametrize("solver,ts,mode", [("euler", "stream40", "fp32"), ("midpoint", "shipped", "fp32"), ("rk4", "shipped", "fp32"), ("euler", "stream40", "bf16x3")])
def test_persistent_segments_equal_launch_per_layer_bitwise(tmp_path, solver, ts, mode):
    a = _run(solver, ts, mode, 1, str(tmp_path / "p.pt"))
    b = _run(solver, ts, mode, 0, str(tmp_path / "l.pt"))
    assert len(a) == len(b) == 6
    for x, y in zip(a, b):
        assert torch.isfinite(x).all()
        assert torch.equal(x, y), (solver, ts, mode, float((x - y).abs().max()))


ERR_SCRIPT = r"""
import sys, ctypes, torch
sys.path.insert(0, %r); sys.path.insert(0, %r)
from util import build_pair, cases, hashfill
from streamingflow_amd import schedule as S, _lib, runtime
C, h, w = 64, 50, 50
cts, lts, tts, dt = cases.timeset("shipped")
net, _ = build_pair(C, "euler", True, True, dt)
ode = net.gru_ode
ode.use_graph = False
times, _ = S.merge_observations(cts[0].tolist(), lts[0].tolist())
sc = S.build_schedule(times, tts[0].tolist(), dt, True, "euler")
hx = (hashfill.normal("pshx0", (len(times), h, w, C), 61) * 0.5).cuda()
eps = hashfill.normal("pseps0", (sc.n_draws, h, w, C), 62).cuda()
a, fa = ode.rollout_nhwc(hx, sc, eps)
errs = _lib.lib().sf_flow_errors(runtime.stream_ptr(hx.device))
print("ERRS", errs, int(torch.isnan(a).all()), int(torch.isnan(fa).all()), int(torch.isfinite(a).all()))
""" % (ROOT, os.path.join(ROOT, "tests"))


def test_flow_timeouts_are_loud():
    """every dependency wait of the flow kernel is bounded; a wait that gave up must not pass as a result.  With the bound
    at ONE poll waits do time out: the rollout's outputs are NaN and sf_flow_errors reports them.  With the default bound the same
    rollout is healthy: zero errors, finite outputs."""
    def run(timeout):
        env = dict(os.environ)
        env["SF_PERSIST"] = "1"
        if timeout is not None:
            env["SF_FLOW_TIMEOUT"] = str(timeout)
        r = subprocess.run([sys.executable, "-c", ERR_SCRIPT], env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        line = [l for l in r.stdout.splitlines() if l.startswith("ERRS")][-1].split()
        return [int(v) for v in line[1:]]
    errs, nan_a, nan_f, finite = run(1)
    assert errs > 0 and nan_a == 1 and nan_f == 1, (errs, nan_a, nan_f)
    errs, nan_a, nan_f, finite = run(None)
    assert errs == 0 and finite == 1, (errs, finite)
