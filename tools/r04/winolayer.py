"""One Winograd layer, `reps` launches (PMC / trace target).  Usage: python3 tools/r04/winolayer.py <c0> <c1> <cout> <n> <H> <W> <reps>"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    from streamingflow_amd import _lib, packing, runtime
    c0, c1, cout, n, H, W, reps = (int(v) for v in sys.argv[1:8])
    L = _lib.lib()
    dev = torch.device("cuda", 0)
    g = torch.Generator(device="cuda").manual_seed(1)
    a0 = torch.randn((n, H, W, c0), device=dev, generator=g)
    a1 = torch.randn((n, H, W, c1), device=dev, generator=g) if c1 else None
    w = torch.randn((cout, c0 + c1, 3, 3), device=dev, generator=g) * (1.0 / (3.0 * (c0 + c1) ** 0.5))
    out = torch.empty((n, H, W, cout), device=dev)
    pk = packing.Pack(None)
    cw = packing.conv_w(pk, w, c0, c1, act="relu", pad=1)
    _lib.check(L.sf_conv2d_repeat(ctypes.byref(cw), runtime.ptr(a0), runtime.ptr(a1), None, ctypes.c_void_p(out.data_ptr()), n, H, W, 0, reps, None, 0,
                                  runtime.stream_ptr(dev)), "run")
    torch.cuda.synchronize()
    print("done")


if __name__ == "__main__":
    main()
