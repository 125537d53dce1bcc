"""One jump + N Euler ODE steps on one 50x50x64 latent through sf_nnfo_rollout_fwd, `reps` times, eager — the PMC target for the
pipelined single-latent step (tools/r03/final_profile.sh): (counter(N=30) - counter(N=10)) / (20 * reps) = per steady-state step.
Usage: python3 tools/chainrun.py <n_steps> <reps>"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from util import build_pair  # noqa: E402
from chainbench import chain_schedule  # noqa: E402


def main():
    n, reps = int(sys.argv[1]), int(sys.argv[2])
    C, h, w = 64, 50, 50
    net, _ = build_pair(C, "euler", True, True, 0.05)
    ode = net.gru_ode
    ode.use_graph = False      # eager: the counters are read per kernel launch
    sc = chain_schedule(n, "euler")
    hx = torch.randn(1, h, w, C, device="cuda") * 0.5
    e = torch.randn(sc.n_draws, h, w, C, device="cuda")
    torch.cuda.synchronize()
    for _ in range(reps):
        ode.rollout_nhwc(hx, sc, e)
    torch.cuda.synchronize()
    print("done", n, reps)


if __name__ == "__main__":
    main()
