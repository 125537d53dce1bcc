"""Per-workgroup start / duration of ONE launch of an eager single-latent rollout (diagnostic -DSF_STAMP build), by workgroup id range.
Usage: SF_LIB_PATH=build_var/stamp/libsfnative.so python3 tools/r06/stamps_wg_detail.py <n_wgs of the launch to pick> [n_steps]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from util import build_pair  # noqa: E402
from chainbench import chain_schedule  # noqa: E402
from streamingflow_amd import _lib  # noqa: E402


def main():
    want = int(sys.argv[1])
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    C, h, w = 64, 50, 50
    net, _ = build_pair(C, "euler", True, True, 0.05)
    ode = net.gru_ode
    ode.use_graph = False
    sc = chain_schedule(n, "euler")
    hx = torch.randn(1, 1, h, w, C, device="cuda") * 0.5
    e = torch.randn(sc.n_draws, 1, h, w, C, device="cuda")
    for _ in range(3):
        ode.rollout_nhwc(hx, sc, e)
    torch.cuda.synchronize()
    st = torch.zeros((64, 4096, 16), dtype=torch.int64, device="cuda")
    L = _lib.lib()
    _lib.check(L.sf_debug_stamps(st.data_ptr()), "stamps")
    ode.rollout_nhwc(hx, sc, e)
    torch.cuda.synchronize()
    _lib.check(L.sf_debug_stamps(None), "stamps")
    a = st.cpu().numpy().astype(np.float64) * 0.01
    slots = [s for s in range(64) if int((a[s][:, 0] > 0).sum()) == want]
    s = slots[-1]
    t = a[s][:want]
    t0 = t[:, 0].min()
    end = np.nanmax(np.where(t[:, :7] > 0, t[:, :7], np.nan), axis=1)
    for lo, hi in ((0, 40), (40, 200), (200, 240), (240, 256), (256, 264), (264, 272), (272, 280)):
        if lo >= want:
            break
        hi = min(hi, want)
        print(f"ids {lo:3d}-{hi - 1:3d}: start +{(t[lo:hi, 0] - t0).min():6.2f} .. +{(t[lo:hi, 0] - t0).max():6.2f} us, duration {np.median(end[lo:hi] - t[lo:hi, 0]):6.2f} (max {np.max(end[lo:hi] - t[lo:hi, 0]):6.2f}), end +{(end[lo:hi] - t0).max():6.2f}")


if __name__ == "__main__":
    main()
