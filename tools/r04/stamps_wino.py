"""In-kernel time stamps of ONE Winograd launch (diagnostic build: tools/build_variant.sh stamp -DSF_STAMP).
Per workgroup (the first 4096 of the launch): entry, end of prologue, end of the stage loop, end of the epilogue, stores drained,
plus HW_ID / XCC_ID -> which workgroups share a CU and how their phases overlap.
Usage: SF_LIB_PATH=build_var/stamp/libsfnative.so python3 tools/r04/stamps_wino.py [cout cin n H W dil]"""
import ctypes
import os
import sys
from collections import defaultdict

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from streamingflow_amd import _lib, packing, runtime  # noqa: E402


def main():
    a = [int(x) for x in sys.argv[1:]]
    cout, cin, n, H, W = a[:5] if len(a) >= 5 else (128, 128, 32, 200, 200)
    dil = a[5] if len(a) > 5 else 1
    add = a[6] if len(a) > 6 else 0
    dev = torch.device("cuda", 0)
    w = torch.randn((cout, cin, 3, 3), device=dev) * 0.03
    pk = packing.Pack(None)
    cw = packing.conv_w(pk, w, cin, 0, act="relu", dil=dil, pad=dil)
    x = torch.randn((n, H, W, cin), device=dev)
    res = torch.randn((n, H, W, cout), device=dev) if add else None
    out = torch.empty((n, H, W, cout), device=dev)
    L = _lib.lib()
    st = torch.zeros((64, 4096, 16), dtype=torch.int64, device=dev)

    def conv():
        _lib.check(L.sf_conv2d_fwd(ctypes.byref(cw), runtime.ptr(x), None, runtime.ptr(res), runtime.ptr(out), n, H, W, 0, runtime.stream_ptr(dev)), "conv")
    for _ in range(3):
        conv()
    torch.cuda.synchronize()
    _lib.check(L.sf_debug_stamps(st.data_ptr()), "stamps")
    conv()
    torch.cuda.synchronize()
    _lib.check(L.sf_debug_stamps(None), "stamps")
    raw = st.cpu().numpy()[0]
    m = raw[:, 0] > 0
    raw = raw[m]
    t = raw[:, :5].astype(np.float64) * 0.01          # us
    t -= t[:, 0].min()
    nkc = int(np.median(raw[:, 10]))
    pro, loop, epi, drain = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2], t[:, 4] - t[:, 3]
    print(f"{cout} <- {cin}, {n} x {H}x{W}, dilation {dil}, residual {bool(add)}: {len(t)} stamped workgroups, {nkc} chunks")
    # steady state: workgroups that started after the first dispatch round
    late = t[:, 0] > np.percentile(t[:, 0], 30)
    for name, v in (("prologue", pro), ("stage loop", loop), ("epilogue", epi), ("store drain", drain), ("total", t[:, 4] - t[:, 0])):
        print(f"  {name:12s} median {np.median(v[late]):7.2f} us   p10 {np.percentile(v[late], 10):7.2f}   p90 {np.percentile(v[late], 90):7.2f}")
    tt = raw[:, :16].astype(np.float64) * 0.01
    def seg(a, b):
        v = (tt[:, b] - tt[:, a])[late]
        return f"{np.median(v):6.2f}"
    print(f"  prologue detail: kernel arguments + block decode + buffer descriptors {seg(0, 14)}  patch lane offsets {seg(14, 15)}  U lane offsets, fragment offsets, DMA issue {seg(15, 11)}")
    print(f"  prologue: index math + DMA issue {seg(0, 11)}  wait for the patch + barrier {seg(11, 12)}  transform {seg(12, 13)}  rest (U(0) wait, barrier, accumulator init, first fragments) {seg(13, 1)}")
    print(f"  epilogue: operand requests + column-0 transform {seg(2, 5)}  column-0 arithmetic + stores {seg(5, 6)}  column-1 transform {seg(6, 7)}  column-1 arithmetic + stores {seg(7, 3)}")
    print(f"  stage loop per chunk {np.median(loop[late]) / nkc:6.2f} us; MFMA floor of a chunk with the pipe to itself {4096 / 2400:.2f} us, shared by two workgroups {8192 / 2400:.2f} us")
    hw, xcc = raw[:, 8], raw[:, 9] & 0xf
    cu = (hw >> 8) & 0xf
    sh = (hw >> 12) & 1
    se = (hw >> 13) & 7
    groups = defaultdict(list)
    for i in range(len(t)):
        groups[(int(xcc[i]), int(se[i]), int(sh[i]), int(cu[i]))].append(i)
    print(f"  distinct (xcc, se, sh, cu): {len(groups)}")
    # overlap structure per CU: fraction of a workgroup's stage loop during which its partner is NOT in its stage loop
    alone, offs = [], []
    for key, idx in groups.items():
        idx = sorted(idx, key=lambda i: t[i, 0])
        for i in idx:
            if not late[i]:
                continue
            a0, a1 = t[i, 1], t[i, 2]
            cov = 0.0
            for j in idx:
                if j == i:
                    continue
                lo, hi = max(a0, t[j, 1]), min(a1, t[j, 2])
                if hi > lo:
                    cov += hi - lo
            alone.append(1.0 - min(cov / max(a1 - a0, 1e-9), 1.0))
        for a_, b_ in zip(idx[:-1], idx[1:]):
            offs.append(t[b_, 0] - t[a_, 0])
    print(f"  share of a workgroup's stage loop with no partner in ITS stage loop: median {np.median(alone):.3f}, mean {np.mean(alone):.3f}")
    k0 = sorted(groups.items(), key=lambda kv: -len(kv[1]))[0]
    print(f"  one CU {k0[0]}: (entry, loop start, loop end, exit) of its workgroups:")
    for i in sorted(k0[1], key=lambda i: t[i, 0])[:10]:
        print("     " + "  ".join(f"{v:8.2f}" for v in (t[i, 0], t[i, 1], t[i, 2], t[i, 4])))


if __name__ == "__main__":
    main()
