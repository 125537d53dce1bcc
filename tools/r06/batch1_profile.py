"""Where the single-sample forward (BASELINE config 2 at the reference's batch size, evaluate.py:46) spends its time: the library's per-launch
hipEvent profiler over one forward, by kernel family, and the wall time of the module-default forward.
Usage: python3 tools/r06/batch1_profile.py [batch]"""
import ctypes
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import streamingflow_amd as sfa  # noqa: E402
from streamingflow_amd import _lib  # noqa: E402
from workloads import synthetic as cases  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    C, H, W = 64, 200, 200
    dev = torch.device("cuda", 0)
    cts, lts, tts, dt = cases.timeset("shipped")
    cfg = cases.make_cfg(C, impute=True, solver="euler", variable=True)
    net = sfa.FuturePredictionODE(C, C, 4, cfg, n_gru_blocks=2, n_res_layers=1, delta_t=dt).eval()
    net.load_state_dict(cases.fpode_state_dict(net.state_dict()))
    net = net.to(dev)
    cams, lids = zip(*[cases.bev_inputs(C, H, W, cts.shape[1], lts.shape[1], seed=i) for i in range(B)])
    cam, lid = torch.cat(cams, 0).to(dev), torch.cat(lids, 0).to(dev)
    x = cases.present_input(cam, lid)
    args = (x, cam, lid, cts.repeat(B, 1), lts.repeat(B, 1), tts.repeat(B, 1))

    def wall(n=10):
        for _ in range(3):
            net(*args)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            net(*args)
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / n
    ms = wall()
    L = _lib.lib()
    net.gru_ode.use_graph = False
    L.sf_prof_enable(1)
    net(*args)
    torch.cuda.synchronize()
    NK = _lib.SF_PROF_KEYS
    calls = (ctypes.c_int32 * NK)(); pms = (ctypes.c_double * NK)(); pfl = (ctypes.c_double * NK)(); pby = (ctypes.c_double * NK)()
    L.sf_prof_collect(calls, pms, pfl, pby)
    L.sf_prof_enable(0)
    net.gru_ode.use_graph = None
    fam = sorted(((pms[i], _lib.KERNEL_NAMES.get(i, str(i)), calls[i], pfl[i]) for i in range(NK) if calls[i]), reverse=True)
    print(json.dumps({"batch": B, "ms_per_forward_module_defaults": ms, "conv_ms_by_profiler": sum(pms)}))
    for t, name, n, fl in fam:
        print(f"{t:8.3f} ms  {n:4d} launches  {fl / (t * 1e-3) / 1e12 if t else 0:7.1f} TFLOP/s  {name}")


if __name__ == "__main__":
    main()
