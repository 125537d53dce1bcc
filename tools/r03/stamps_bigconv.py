"""In-kernel time stamps of ONE large convolution on the LDS-DMA kernel (diagnostic build: tools/build_variant.sh stamp -DSF_STAMP).
Usage: SF_LIB_PATH=build_var/stamp/libsfnative.so python3 tools/r03/stamps_bigconv.py [cout cin n H W]"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from streamingflow_amd import _lib, packing, runtime  # noqa: E402

NAMES = ["entry", "prologue", "1st chunk", "K loop", "hand-off", "epilogue", "drain"]


def main():
    cout, cin, n, H, W = (int(x) for x in sys.argv[1:6]) if len(sys.argv) > 5 else (128, 128, 32, 200, 200)
    dev = torch.device("cuda", 0)
    w = torch.randn((cout, cin, 3, 3), device=dev) * 0.03
    pk = packing.Pack(None)
    cw = packing.conv_w(pk, w, cin, 0, act="relu")
    x = torch.randn((n, H, W, cin), device=dev)
    out = torch.empty((n, H, W, cout), device=dev)
    L = _lib.lib()
    st = torch.zeros((64, 4096, 16), dtype=torch.int64, device=dev)

    def conv():
        _lib.check(L.sf_conv2d_fwd(ctypes.byref(cw), runtime.ptr(x), None, None, runtime.ptr(out), n, H, W, 0, runtime.stream_ptr(dev)), "conv")
    for _ in range(3):
        conv()
    torch.cuda.synchronize()
    _lib.check(L.sf_debug_stamps(st.data_ptr()), "stamps")
    conv()
    torch.cuda.synchronize()
    _lib.check(L.sf_debug_stamps(None), "stamps")
    raw = st.cpu().numpy().astype(np.float64)
    t = raw[0] * 0.01
    m = t[:, 0] > 0
    t = t[m]
    seg = np.diff(t[:, :7], axis=1)
    tot = t[:, 6] - t[:, 0]
    r = raw[0][m]
    nch = np.median(r[:, 12])
    print(f"{len(t)} stamped workgroups (the first 4096 of the launch), {nch:.0f} chunks of 32: median per-WG us: " +
          "  ".join(f"{NAMES[k + 1]} {np.median(seg[:, k]):6.2f}" for k in range(6)) + f" | total {np.median(tot):7.2f}")
    print(f"K loop cycles per chunk (wave 0): {np.median(r[:, 8]) / nch:6.0f}  of which DMA issue {np.median(r[:, 9]) / nch:5.0f}  vmcnt+lgkm wait {np.median(r[:, 10]) / nch:5.0f}  barrier {np.median(r[:, 11]) / nch:5.0f}")
    # concurrency among the stamped workgroups (they are the first 4096 of the launch: the early part of the kernel)
    ev = sorted([(a, 1) for a in t[:, 0]] + [(b, -1) for b in t[:, 6]])
    cur, peak, area, last = 0, 0, 0.0, ev[0][0]
    t_lo, t_hi = np.percentile(t[:, 0], 20), np.percentile(t[:, 0], 70)
    for tt, d in ev:
        if t_lo <= last and tt <= t_hi:
            area += cur * (tt - last)
        cur += d; peak = max(peak, cur); last = tt
    print(f"workgroups in flight: peak {peak}, mean over the middle of the stamped range {area / max(t_hi - t_lo, 1e-9):.0f} (256 CUs)")
    flops = 2.0 * 128 * 128 * nch * 32
    print(f"K loop share of the workgroup's life: {np.median(seg[:, 2]) / np.median(tot):.3f}; MFMA floor of the loop at 2.4 GHz: {nch * 4096 / 2 / 2400:.1f} us (2 workgroups per CU share the pipe)")


if __name__ == "__main__":
    main()
