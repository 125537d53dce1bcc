"""In-kernel time stamps (diagnostic build: tools/build_variant.sh stamp -DSF_STAMP) of the small-P launches of an eager single-latent rollout
(1 jump + N Euler steps, the carried form: 9 launches per step).  Prints the launches of the last 64 slots in launch order.
Usage: SF_LIB_PATH=build_var/stamp/libsfnative.so python3 tools/r06/stamps_rollout.py [n_steps]"""
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

NAMES = ["prologue", "1st data", "K loop", "hand-off", "epilogue", "drain"]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
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
    raw = st.cpu().numpy().astype(np.float64)
    a = raw * 0.01      # us
    starts = [(a[s][:, 0][a[s][:, 0] > 0].min(), s) for s in range(64) if (a[s][:, 0] > 0).any()]
    starts.sort()
    prev_end = None
    for t_first, slot in starts:
        t = a[slot]
        t = t[t[:, 0] > 0]
        t0 = t[:, 0].min()
        pts = np.where(t[:, :7] > 0, t[:, :7], np.nan)
        seg = np.diff(pts, axis=1)
        end = np.nanmax(pts)
        gap = (t0 - prev_end) if prev_end is not None else 0.0
        extra = ""
        if (t[:, 9] > 0).any():      # Winograd form: stamp 9 = output transform done
            k = t[:, 9] > 0
            extra = (f" | after the loop: operands issued +{np.nanmedian(t[k, 10] - t[k, 3]):4.2f}, barrier +{np.nanmedian(t[k, 14] - t[k, 3]):4.2f}, M stored + barrier +{np.nanmedian(t[k, 15] - t[k, 3]):4.2f}, "
                     f"out-transform +{np.nanmedian(t[k, 9] - t[k, 3]):4.2f}")
        if (t[:, 11] > 0).any():      # SE gate in the prologue: rows summed / first barrier / second barrier, relative to stamp 0
            k = t[:, 11] > 0
            extra += f" | SE gate: rows summed +{np.nanmedian(t[k, 11] - t[k, 0]):4.2f}, barrier +{np.nanmedian(t[k, 12] - t[k, 0]):4.2f}, hidden units + barrier +{np.nanmedian(t[k, 13] - t[k, 0]):4.2f}"
        print(f"slot {slot:2d}: {len(t):4d} WGs gap {gap:5.2f} skew {t[:, 0].max() - t0:5.2f} span {end - t0:6.2f} us | " +
              "  ".join(f"{NAMES[k]} {np.nanmedian(seg[:, k]):5.2f}" for k in range(6)) + f" | max WG {np.nanmax(np.nanmax(pts, axis=1) - t[:, 0]):5.2f}" + extra)
        prev_end = end


if __name__ == "__main__":
    main()
