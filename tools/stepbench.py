"""Run N fused GRU-ODE Euler steps (sf_ode_step_fwd) and nothing else — the target of tools/pmc_step.sh.
Usage: python3 tools/stepbench.py <batch> <h> <w> <n_steps>"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from util import build_pair  # noqa: E402
from streamingflow_amd import _lib, runtime, schedule as S  # noqa: E402


def main():
    B, h, w, n = (int(x) for x in sys.argv[1:5])
    C = 64
    net, _ = build_pair(C)
    ode = net.gru_ode
    dev = torch.device("cuda", 0)
    s = torch.randn((B, h, w, C), device=dev) * 0.5
    p = torch.randn((B, h, w, C), device=dev) * 0.5
    e = torch.randn((1, B, h, w, C), device=dev)
    so, po = torch.empty_like(s), torch.empty_like(p)
    coef = torch.from_numpy(S.Schedule(dts=[0.05]).coef_array()).to(dev)
    L = _lib.lib()
    ws = runtime.workspace(L.sf_ode_step_ws_bytes(C, B, h, w), dev)
    pr = runtime.ptr
    torch.cuda.synchronize()
    for _ in range(n):
        _lib.check(L.sf_ode_step_fwd(ode.gru_c.packed().struct, ode.p_model.packed().struct, _lib.SOLVER["euler"], 1, pr(s), pr(p),
                                     pr(coef), pr(e), pr(so), pr(po), B, h, w, pr(ws), ws.numel() * 4, runtime.stream_ptr(dev)),
                   "ode_step")
    torch.cuda.synchronize()
    print("done", n)


if __name__ == "__main__":
    main()
