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
    env["SF_WINO_SP"] = "0"      # the flow kernel keeps the direct form of every 3x3 layer: bitwise against the launch path in the same form
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
from streamingflow_amd import schedule as S, _lib, runtime, packing
packing.set_persistent_flow(True, check=False)      # unchecked: a timeout shows as NaN (the checked mode would re-run the rollout)
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


# ---- the flow kernel against the ORACLE (the bitwise test above compares it with its sibling only) --------------------
ORACLE_SCRIPT = r"""
import sys, torch
sys.path.insert(0, %r); sys.path.insert(0, %r)
from util import build_pair, cases, hashfill, oracle_rollout, maxabs
from streamingflow_amd import schedule as S, packing
solver = sys.argv[1]
C, h, w = 64, 50, 50
cts, lts, tts, dt = cases.timeset("stream40")
net, sd = build_pair(C, solver, True, True, dt)
ode = net.gru_ode
times, _ = S.merge_observations(cts[0].tolist(), lts[0].tolist())
sc = S.build_schedule(times, tts[0].tolist(), dt, True, solver)
hx = hashfill.normal("orhx", (len(times), h, w, C), 71) * 0.5
eps = hashfill.normal("oreps", (sc.n_draws, h, w, C), 72)
ref, ref_final = oracle_rollout(sd, sc, hx, eps, solver)
worst = 0.0
for use_graph in (False, True):
    ode.use_graph = use_graph
    a, fa = ode.rollout_nhwc(hx.cuda(), sc, eps.cuda())
    worst = max(worst, maxabs(a, ref), maxabs(fa, ref_final))
print("ORACLE", worst, int(packing.flow_active()), packing.FLOW_FALLBACKS[0])
""" % (ROOT, os.path.join(ROOT, "tests"))


@pytest.mark.parametrize("persist", [1, 0, "fork7"])
@pytest.mark.parametrize("solver", ["euler", "rk4"])
def test_c64_stream40_rollout_vs_oracle(solver, persist):
    """One C = 64, 50x50 rollout over the 46-step streaming schedule (8 jumps + 46 steps chained, eager and as a replayed hipGraph)
    against the oracle's latent-level composition of its own jump / step / infer_state: <= 1e-3, the north-star tolerance.
    persist = 1: on the persistent flow kernel, in a fresh process (and no wait may have timed out: no fallback counted);
    persist = 0: the default launch path (Winograd form of the 3x3 layers)."""
    env = dict(os.environ)
    if persist == "fork7":      # opt-in form: the r2 half of the next cell's 7x7 on a forked stream beside infer_state (csrc/api.hip: fork_run)
        env["SF_FORK7"] = "1"
        persist = 0
    env["SF_PERSIST"] = str(persist)
    r = subprocess.run([sys.executable, "-c", ORACLE_SCRIPT, solver], env=env, capture_output=True, text=True, timeout=1800)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("ORACLE")][-1].split()
    worst, active, fallbacks = float(line[1]), int(line[2]), int(line[3])
    assert active == persist and fallbacks == 0
    assert worst <= 1e-3, worst


def test_golden_and_config5_statistics_on_the_flow_kernel():
    """The reference-generated fixture `c8_16_shipped` and the full-size config-5 statistics (C = 64, BEV 200x200, 46 steps, 43 frames,
    euler: from the REAL reference) with SF_PERSIST=1: the very tests of test_gpu_forward.py / test_gpu_configs.py, re-run in a child
    pytest whose rollouts go through the flow kernel."""
    env = dict(os.environ)
    env["SF_PERSIST"] = "1"
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-m", "gpu", "-p", "no:cacheprovider",
                        os.path.join(ROOT, "tests", "test_gpu_forward.py"), os.path.join(ROOT, "tests", "test_gpu_configs.py"),
                        "-k", "(test_fpode_golden and c8_16_shipped) or (test_config5_full_size and euler)"],
                       env=env, capture_output=True, text=True, timeout=3000, cwd=ROOT)
    tail = r.stdout[-1500:] + r.stderr[-1500:]
    assert r.returncode == 0, tail
    import re
    m = re.search(r"(\d+) passed", r.stdout)
    assert m and int(m.group(1)) >= 2 and "failed" not in r.stdout, tail


FALLBACK_SCRIPT = r"""
import sys, torch
sys.path.insert(0, %r); sys.path.insert(0, %r)
from util import build_pair, cases, hashfill
from streamingflow_amd import schedule as S, packing
C, h, w = 64, 50, 50
cts, lts, tts, dt = cases.timeset("shipped")
net, _ = build_pair(C, "euler", True, True, dt)
ode = net.gru_ode
times, _ = S.merge_observations(cts[0].tolist(), lts[0].tolist())
sc = S.build_schedule(times, tts[0].tolist(), dt, True, "euler")
hx = (hashfill.normal("pshx0", (len(times), h, w, C), 61) * 0.5).cuda()
eps = hashfill.normal("pseps0", (sc.n_draws, h, w, C), 62).cuda()
packing.set_persistent_flow(False)
ode.use_graph = False
ref, ref_final = ode.rollout_nhwc(hx, sc, eps)
packing.set_persistent_flow(True)            # checked mode (the default of the switch)
res = []
for use_graph in (False, True, True):
    ode.use_graph = use_graph
    a, fa = ode.rollout_nhwc(hx, sc, eps)
    res.append(int(torch.equal(a, ref) and torch.equal(fa, ref_final)))
print("FALLBACK", packing.FLOW_FALLBACKS[0], *res)
""" % (ROOT, os.path.join(ROOT, "tests"))


def test_flow_timeout_falls_back_to_the_launch_path_in_process():
    """set_persistent_flow(True) is checked: with the wait bound at ONE poll every persistent rollout times out — and every one is run
    again on the launch-per-layer path in the same process (eager and graph replays alike), counted, and returns that path's bits."""
    env = dict(os.environ)
    env["SF_FLOW_TIMEOUT"] = "1"
    env["SF_WINO_SP"] = "0"
    r = subprocess.run([sys.executable, "-c", FALLBACK_SCRIPT], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("FALLBACK")][-1].split()
    assert int(line[1]) == 3 and line[2:] == ["1", "1", "1"], line
