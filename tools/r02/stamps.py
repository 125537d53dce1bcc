"""In-kernel time stamps of one Euler ode_step (diagnostic build: tools/build_variant.sh stamp -DSF_STAMP).
Usage: SF_LIB_PATH=build_var/stamp/libsfnative.so python3 tools/r02/stamps.py <batch> <h> <w>"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from util import build_pair  # noqa: E402
from streamingflow_amd import _lib, runtime, schedule as S  # noqa: E402

NAMES = ["entry", "prologue", "1st chunk", "K loop", "hand-off", "epilogue", "drain"]


def main():
    B, h, w = (int(x) for x in sys.argv[1:4])
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
    st = torch.zeros((64, 4096, 16), dtype=torch.int64, device=dev)

    def step():
        _lib.check(L.sf_ode_step_fwd(ode.gru_c.packed().struct, ode.p_model.packed().struct, _lib.SOLVER["euler"], 1, pr(s), pr(p),
                                     pr(coef), pr(e), pr(so), pr(po), B, h, w, pr(ws), ws.numel() * 4, runtime.stream_ptr(dev)), "ode_step")
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    _lib.check(L.sf_debug_stamps(st.data_ptr()), "stamps")
    step()
    step()
    torch.cuda.synchronize()
    _lib.check(L.sf_debug_stamps(None), "stamps")
    raw = st.cpu().numpy().astype(np.float64)
    a = raw * 0.01      # us
    prev_end = None
    # conv launches of the second step: slots 11..21 (11 conv launches per step; the SE kernels are not stamped)
    nl = int(os.environ.get("NL", "11"))
    for slot in range(nl, 2 * nl):
        t = a[slot]
        m = t[:, 0] > 0
        t = t[m]
        if not len(t):
            continue
        t0 = t[:, 0].min()
        gap = (t0 - prev_end) if prev_end is not None else 0.0
        seg = np.diff(t[:, :7], axis=1)
        seg = np.where(t[:, 1:7] > 0, seg, np.nan)
        end = np.nanmax(np.where(t[:, :7] > 0, t[:, :7], np.nan))
        print(f"launch {slot - nl:2d}: {len(t):5d} WGs, gap {gap:5.2f} us, start skew {t[:, 0].max() - t0:5.2f}, span {end - t0:6.2f} us | median per-WG segment us: " +
              "  ".join(f"{NAMES[k + 1]} {np.nanmedian(seg[:, k]):5.2f}" for k in range(6)) + f" | max WG total {np.nanmax(np.nanmax(np.where(t[:, :7] > 0, t[:, :7], np.nan), axis=1) - t[:, 0]):6.2f}")
        prev_end = end
        r = raw[slot][m]
        if r[:, 12].max() > 0:
            k = r[:, 12] > 0
            nch = np.median(r[k, 12])
            if r[k, 13].max() > 0:
                print(f"           consumer wave 0: {np.median(r[k, 8]) / max(nch, 1):6.0f} cycles / chunk of which barrier wait {np.median(r[k, 11]) / max(nch, 1):5.0f} | loader wave 8 per chunk: issue {np.median(r[k, 13]) / max(nch, 1):5.0f}  vmcnt wait {np.median(r[k, 14]) / max(nch, 1):5.0f}  barrier wait {np.median(r[k, 15]) / max(nch, 1):5.0f}   ({nch:.0f} chunks of 64)")
                continue
            print(f"           K loop cycles (wave 0, median): total {np.median(r[k, 8]):8.0f} for {nch:.0f} chunks = {np.median(r[k, 8]) / max(nch, 1):6.0f} / chunk | DMA issue {np.median(r[k, 9]) / max(nch, 1):5.0f}  vmcnt+lgkm wait {np.median(r[k, 10]) / max(nch, 1):5.0f}  barrier {np.median(r[k, 11]) / max(nch, 1):5.0f}  (rest: fragment reads + MFMA)")


if __name__ == "__main__":
    main()
