"""64 -> 64 Winograd layer at 200x200 for several image counts: does the per-image time depend on the launch size?  (The encoder's n = 256
launches measured 0.56-0.59 of the peak against 0.65 for the same layer at n = 224.)  Usage: python3 tools/r05/wino_n_sweep.py"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    from streamingflow_amd import _lib, packing, runtime
    L = _lib.lib()
    dev = torch.device("cuda", 0)
    e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
    L.sf_event_create(ctypes.byref(e0)); L.sf_event_create(ctypes.byref(e1))
    ms = ctypes.c_float()
    for (c0, cout, H, W) in ((64, 64, 200, 200), (128, 128, 200, 200)):
        for n in (64, 128, 192, 224, 256, 288):
            a0 = torch.randn((n, H, W, c0), device=dev)
            w = torch.randn((cout, c0, 3, 3), device=dev) * 0.04
            out = torch.empty((n, H, W, cout), device=dev)
            pk = packing.Pack(None)
            cw = packing.conv_w(pk, w, c0, 0, act="lrelu", pad=1)
            args = (ctypes.byref(cw), runtime.ptr(a0), None, None, ctypes.c_void_p(out.data_ptr()), n, H, W, 0)
            _lib.check(L.sf_conv2d_repeat(*args, 2, None, 0, runtime.stream_ptr(dev)), "warm")
            torch.cuda.synchronize()
            L.sf_event_record(e0, runtime.stream_ptr(dev))
            _lib.check(L.sf_conv2d_repeat(*args, 5, None, 0, runtime.stream_ptr(dev)), "timed")
            L.sf_event_record(e1, runtime.stream_ptr(dev))
            torch.cuda.synchronize()
            L.sf_event_elapsed_ms(e0, e1, ctypes.byref(ms))
            t = ms.value / 5
            fl = 2.0 * 16 * n * ((H + 1) // 2) * ((W + 1) // 2) * cout * c0
            print(f"{c0}->{cout} n={n:4d}: {t:8.3f} ms  {t / n * 1e3:7.2f} us per image  {fl / t * 1e-9 / 157.3:.3f} of peak", flush=True)
            del a0, out


if __name__ == "__main__":
    main()
