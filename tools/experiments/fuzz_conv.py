"""One-off fuzz of sf_conv2d_ex_fwd over layer shapes that reach every conv kernel of the library (LDS-DMA tiles of all
families, split-K, direct-fragment, register-staged), against torch's own convolution on the same device (fp32, MIOpen) —
a second opinion next to the CPU-checked tests/test_gpu_conv_random.py.  Usage (GPU box): python tools/experiments/fuzz_conv.py [n]"""
import ctypes
import os
import random
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from streamingflow_amd import _lib, packing, runtime  # noqa: E402


SEED0 = 7000


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    global SEED0
    SEED0 = int(sys.argv[2]) if len(sys.argv) > 2 else 7000
    torch.backends.cudnn.allow_tf32 = False
    L = _lib.lib()
    ws = runtime.workspace(L.sf_conv2d_ex_ws_bytes(), "cuda")
    worst, bad = 0.0, 0
    for i in range(n_cases):
        r = random.Random(SEED0 + i)
        k = r.choice([1, 3, 3, 3, 5, 7])
        stride = r.choice([1, 1, 1, 2])
        dil = r.choice([1, 1, 2, 6, 12]) if k == 3 and stride == 1 else 1
        c0 = r.choice([16, 32, 64, 96, 128, 40])
        c1 = r.choice([0, 0, 32, 64]) if c0 % 32 == 0 else r.choice([0, 8])
        cout = r.choice([16, 32, 64, 128, 192, 256, 24])
        up = r.choice([0, 0, 0, 1])
        big = r.random() < 0.5
        if big:
            n, H, W = r.choice([(2, 200, 200), (7, 200, 200), (1, 300, 333), (40, 50, 50), (16, 100, 90), (600, 13, 17)])
        else:
            n, H, W = r.choice([(1, 50, 50), (2, 50, 50), (4, 50, 50), (1, 64, 37), (3, 20, 21), (8, 50, 50)])
        if up and n * H * W * 4 > 1200000:
            up = 0
        pad = r.choice([dil * (k - 1) // 2, 0]) if k > 1 else 0
        Hl, Wl = H << up, W << up
        if (Hl + 2 * pad - dil * (k - 1) - 1) // stride + 1 < 1 or (Wl + 2 * pad - dil * (k - 1) - 1) // stride + 1 < 1:
            pad = dil * (k - 1) // 2
        g = torch.Generator(device="cuda").manual_seed(i)
        x0 = torch.randn((n, H, W, c0), device="cuda", generator=g)
        x1 = torch.randn((n, H, W, c1), device="cuda", generator=g) if c1 else None
        w = torch.randn((cout, c0 + c1, k, k), device="cuda", generator=g) * (2.0 / ((c0 + c1) * k * k)) ** 0.5
        b = torch.randn((cout,), device="cuda", generator=g)
        act = r.choice(["none", "relu", "lrelu"])
        pk = packing.Pack(None)
        cw = packing.conv_w(pk, w, c0, c1, None, b, act, dil=dil, stride=stride, pad=pad)
        xin = torch.cat([x0, x1], -1) if c1 else x0
        xin = xin.permute(0, 3, 1, 2)
        if up:
            xin = F.interpolate(xin, scale_factor=2, mode="nearest")
        y = F.conv2d(xin.double(), w.double(), b.double(), stride, pad, dil)      # fp64 on the GPU: an exact reference
        y = {"none": lambda t: t, "relu": F.relu, "lrelu": lambda t: F.leaky_relu(t, 0.1)}[act](y)
        Ho, Wo = y.shape[-2:]
        out = torch.empty((n, Ho, Wo, cout), device="cuda")
        _lib.check(L.sf_conv2d_ex_fwd(ctypes.byref(cw), runtime.ptr(x0), c0, runtime.ptr(x1), c1, None, cout, 0,
                                      ctypes.c_void_p(out.data_ptr()), cout, 0, n, H, W, up, runtime.ptr(ws), ws.numel() * 4,
                                      runtime.stream_ptr()), "conv2d_ex")
        err = float((out.permute(0, 3, 1, 2).double() - y).abs().max())
        worst = max(worst, err)
        if err > 2e-4:
            bad += 1
            print("MISMATCH", i, dict(k=k, stride=stride, dil=dil, c0=c0, c1=c1, cout=cout, up=up, n=n, H=H, W=W, pad=pad, act=act), err)
    print(f"{n_cases} cases, {bad} mismatches, worst max-abs error {worst:.3e}")


if __name__ == "__main__":
    main()
