"""us per Euler ode_step (sf_ode_step_fwd, eager, hipEvents on the launch stream).  Usage: python3 tools/r02/steptime.py [B h w]..."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from util import build_pair  # noqa: E402
from streamingflow_amd import _lib, runtime, schedule as S  # noqa: E402


def run(B, h, w, n=100, C=64):
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

    def step():
        _lib.check(L.sf_ode_step_fwd(ode.gru_c.packed().struct, ode.p_model.packed().struct, _lib.SOLVER["euler"], 1, pr(s), pr(p),
                                     pr(coef), pr(e), pr(so), pr(po), B, h, w, pr(ws), ws.numel() * 4, runtime.stream_ptr(dev)), "ode_step")
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            step()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / n * 1e3)
    ts.sort()
    fl = 728.0 * C * C * h * w * B
    print(f"ode_step B={B} {h}x{w}: {ts[2]:8.1f} us/step  ({fl / ts[2] / 1e6:6.1f} TFLOP/s = {fl / ts[2] / 1e6 / 157.3 * 100:4.1f} % of the fp32 MFMA peak)", flush=True)


if __name__ == "__main__":
    a = [int(x) for x in sys.argv[1:]]
    cfgs = [a[i:i + 3] for i in range(0, len(a), 3)] or [[1, 50, 50]]
    for B, h, w in cfgs:
        run(B, h, w, 100 if B * h * w < 20000 else 30)
