"""Steady-state time of one GRU-ODE step inside a rollout (the pipelined form): hipGraph replays of rollouts with one
jump + N ODE steps for two values of N, (t(N2) - t(N1)) / (N2 - N1).  Usage: python3 tools/chainbench.py [solver] [N1 N2] [h w]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from util import build_pair  # noqa: E402
from streamingflow_amd import schedule as S  # noqa: E402
from streamingflow_amd._lib import OP_JUMP, OP_STEP  # noqa: E402


def chain_schedule(n, solver, dt=0.05):
    per = S.DRAWS_PER_STEP[solver]
    return S.Schedule(ops=[(OP_JUMP, 0)] + [(OP_STEP, i) for i in range(n)], dts=[dt] * n, sel_nops=[n + 1], n_draws=1 + per * n)


def time_chain(ode, n, solver, h, w, C, reps=20, B=1):
    sc = chain_schedule(n, solver)
    hx = torch.randn(1, B, h, w, C, device="cuda") * 0.5
    e = torch.randn(sc.n_draws, B, h, w, C, device="cuda")
    ode.use_graph = True
    for _ in range(3):
        ode.rollout_nhwc(hx, sc, e)
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        ode.rollout_nhwc(hx, sc, e)
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def main():
    import streamingflow_amd as sfa
    sfa.set_math_mode(os.environ.get("SF_MATH_MODE", "fp32"))
    solver = sys.argv[1] if len(sys.argv) > 1 else "euler"
    n1, n2 = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (10, 30)
    h, w = (int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (50, 50)
    B = int(sys.argv[6]) if len(sys.argv) > 6 else 1
    C = 64
    net, _ = build_pair(C, solver, True, True, 0.05)
    ode = net.gru_ode
    t1, t2 = time_chain(ode, n1, solver, h, w, C, B=B), time_chain(ode, n2, solver, h, w, C, B=B)
    print(f"chain {solver} {B}x{h}x{w} SF_PIPE={os.environ.get('SF_PIPE', 'default')}: {n1} steps {t1:.1f} us, {n2} steps {t2:.1f} us -> {(t2 - t1) / (n2 - n1):.2f} us per step (steady state)", flush=True)


if __name__ == "__main__":
    main()
