"""Winograd F(2x2, 3x3) go / no-go: the 3x3 layers of the batched head, direct form (conv_glds 128x128 tiles)
against csrc/conv_wino.hip, same inputs, hipEvent-timed back-to-back launches (sf_conv2d_repeat).  One JSON object.
Usage: python3 tools/r04/winobench.py [reps]"""
import ctypes
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
PEAK = 157.3


def main():
    from streamingflow_amd import _lib, packing, runtime
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    L = _lib.lib()
    dev = torch.device("cuda", 0)
    e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
    L.sf_event_create(ctypes.byref(e0)); L.sf_event_create(ctypes.byref(e1))
    ms = ctypes.c_float()
    rows = []
    cases = [("DeepLab conv3 / GRU gates 128->128, 224 x 200x200 (7 frames x 32 samples)", 224, 200, 200, 128, 0, 128),
             ("GRU gates cat[64|64]->128, 224 x 200x200", 224, 200, 200, 64, 64, 128),
             ("encoder 64->128, 256 x 100x100", 256, 100, 100, 64, 0, 128),
             ("128->128, 256 x 100x100", 256, 100, 100, 128, 0, 128),
             ("128->128, 8 x 200x200", 8, 200, 200, 128, 0, 128),
             ("decoder / encoder 64->64, 224 x 200x200", 224, 200, 200, 64, 0, 64),
             ("GRU candidate cat[64|64]->64, 224 x 200x200", 224, 200, 200, 64, 64, 64),
             ("128->64, 256 x 100x100", 256, 100, 100, 128, 0, 64),
             ("ASPP branch 64->128 dilation 12, 224 x 200x200", 224, 200, 200, 64, 0, 128, 12),
             ("ASPP branch 64->128 dilation 24, 224 x 200x200", 224, 200, 200, 64, 0, 128, 24),
             ("ASPP branch 64->128 dilation 36, 224 x 200x200", 224, 200, 200, 64, 0, 128, 36)]
    only = os.environ.get("WINOBENCH_ONLY")
    for case in cases:
        name, n, H, W, c0, c1, cout = case[:7]
        dil = case[7] if len(case) > 7 else 1
        if only and only not in name:
            continue
        g = torch.Generator(device="cuda").manual_seed(1)
        a0 = torch.randn((n, H, W, c0), device=dev, generator=g)
        a1 = torch.randn((n, H, W, c1), device=dev, generator=g) if c1 else None
        w = torch.randn((cout, c0 + c1, 3, 3), device=dev, generator=g) * (1.0 / (3.0 * (c0 + c1) ** 0.5))
        out = torch.empty((n, H, W, cout), device=dev)
        res = {}
        for mode in ("direct", "winograd"):
            packing.set_winograd(mode == "winograd")
            pk = packing.Pack(None)
            cw = packing.conv_w(pk, w, c0, c1, act="relu", dil=dil, pad=dil)
            args = (ctypes.byref(cw), runtime.ptr(a0), runtime.ptr(a1), None, ctypes.c_void_p(out.data_ptr()), n, H, W, 0)
            _lib.check(L.sf_conv2d_repeat(*args, 2, None, 0, runtime.stream_ptr(dev)), "warm")
            torch.cuda.synchronize()
            L.sf_event_record(e0, runtime.stream_ptr(dev))
            _lib.check(L.sf_conv2d_repeat(*args, reps, None, 0, runtime.stream_ptr(dev)), "timed")
            L.sf_event_record(e1, runtime.stream_ptr(dev))
            torch.cuda.synchronize()
            L.sf_event_elapsed_ms(e0, e1, ctypes.byref(ms))
            res[mode] = (ms.value / reps, out.clone() if n <= 16 else out[:2].clone())
        packing.set_winograd(True)
        P = n * H * W
        flops_direct = 2.0 * P * cout * 9 * (c0 + c1)
        def axis_tiles(N):      # conv_wino.hip WnAxis: phases p < N % d have N // d + 1 pixels
            q, r = divmod(N, dil)
            return r * ((q + 2) // 2) + (dil - r) * ((q + 1) // 2)
        flops_wino = 2.0 * 16 * n * axis_tiles(H) * axis_tiles(W) * cout * (c0 + c1)
        td, tw = res["direct"][0], res["winograd"][0]
        rows.append({"layer": name, "direct_ms": td, "winograd_ms": tw, "speedup": td / tw,
                     "direct_tflops": flops_direct / td * 1e-9, "winograd_executed_tflops": flops_wino / tw * 1e-9,
                     "winograd_frac_of_fp32_mfma_peak_executed": flops_wino / tw * 1e-9 / PEAK,
                     "winograd_direct_equivalent_tflops": flops_direct / tw * 1e-9,
                     "max_abs_direct_vs_winograd": float((res["direct"][1] - res["winograd"][1]).abs().max()),
                     "abs_max_of_output": float(res["direct"][1].abs().max()),
                     "winograd_out_checksum": int(res["winograd"][1].view(torch.int32).to(torch.int64).sum())})
        print(json.dumps(rows[-1]), flush=True)
    print(json.dumps({"winobench": rows}))


if __name__ == "__main__":
    main()
