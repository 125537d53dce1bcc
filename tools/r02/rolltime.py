"""ms per single-sample rollout (eager and hipGraph replay) for a timeset.  Usage: python3 tools/r02/rolltime.py [timeset] [solver]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from util import build_pair, cases  # noqa: E402
from streamingflow_amd import schedule as S  # noqa: E402


def main():
    import streamingflow_amd as sfa
    sfa.set_math_mode(os.environ.get("SF_MATH_MODE", "fp32"))
    name = sys.argv[1] if len(sys.argv) > 1 else "shipped"
    solver = sys.argv[2] if len(sys.argv) > 2 else "euler"
    C, h, w = 64, 50, 50
    cts, lts, tts, dt = cases.timeset(name)
    net, _ = build_pair(C, solver, True, True, dt)
    ode = net.gru_ode
    times, _ = S.merge_observations(cts[0].tolist(), lts[0].tolist())
    sc = S.build_schedule(times, tts[0].tolist(), dt, True, solver)
    hx = torch.randn(len(times), h, w, C, device="cuda") * 0.5
    e = torch.randn(sc.n_draws, h, w, C, device="cuda")

    def timeit(reps=10):
        for _ in range(3):
            ode.rollout_nhwc(hx, sc, e)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            ode.rollout_nhwc(hx, sc, e)
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / reps
    t_e = timeit()
    ode.use_graph = True
    t_g = timeit()
    nops = len(sc.ops)
    print(f"rollout {name} {solver}: {sc.n_steps} steps + {sc.n_jumps} jumps: eager {t_e:.3f} ms, hipGraph replay {t_g:.3f} ms = {1e3 * t_g / nops:.1f} us / op", flush=True)


if __name__ == "__main__":
    main()
