"""Per-module timing at BASELINE config 2 sizes (C=64, BEV 200x200) + isolated conv shapes.
Usage (GPU box): python tools/modbench.py > gpurun_out/modbench.txt"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from util import build_pair, cases  # noqa: E402
from streamingflow_amd import _lib, packing, runtime, schedule as S  # noqa: E402


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3   # us


def conv_case(cin, cout, k, n, H, W, dil=1, c1=0):
    x = torch.randn(n, H, W, cin - c1, device="cuda")
    x1 = torch.randn(n, H, W, c1, device="cuda") if c1 else None
    w = torch.randn(cout, cin, k, k, device="cuda") * 0.05
    pk = packing.Pack(None)
    cw = packing.conv_w(pk, w, cin - c1, c1, act="lrelu", dil=dil)
    out = torch.empty(n, H, W, cout, device="cuda")
    L = _lib.lib()
    ws = runtime.workspace(80 << 20, "cuda")
    reps = 50
    fn = lambda: L.sf_conv2d_repeat(ctypes.byref(cw), runtime.ptr(x), runtime.ptr(x1), None, runtime.ptr(out), n, H, W, 0,
                                    reps, runtime.ptr(ws), ws.numel() * 4, runtime.stream_ptr())
    us = timeit(fn, 4, 1) / reps
    fl = 2.0 * n * H * W * cout * cin * k * k
    print(f"conv {cin:4d}->{cout:4d} k{k} d{dil:2d} n{n} {H}x{W}: {us:9.1f} us  {fl / us / 1e6:7.2f} TFLOP/s")


def main():
    C, H, W = 64, 200, 200
    net, _ = build_pair(C)
    ode = net.gru_ode
    h, w = H // 4, W // 4
    s = torch.randn(1, h, w, C, device="cuda") * 0.5
    x = torch.randn(1, h, w, C, device="cuda")
    out = torch.empty_like(s)
    one = torch.ones(1, device="cuda")
    print(f"dual_cell (deriv) 50x50x64      : {timeit(lambda: ode.gru_c.run_nhwc(x, s, out, True, s, one)):9.1f} us  (4.649 GFLOP)")
    L = _lib.lib()
    eps = torch.randn(1, h, w, C, device="cuda")
    p = torch.empty_like(s)
    ws = runtime.workspace(L.sf_infer_state_ws_bytes(C, 1, h, w), "cuda")
    pm = ode.p_model.packed().struct
    f = lambda: L.sf_infer_state_fwd(pm, runtime.ptr(s), runtime.ptr(eps), runtime.ptr(p), None, 1, h, w, runtime.ptr(ws),
                                     ws.numel() * 4, runtime.stream_ptr())
    print(f"infer_state 50x50x64            : {timeit(f):9.1f} us  (2.806 GFLOP)")
    cts, lts, tts, dt = cases.timeset("shipped")
    times, _ = S.merge_observations(cts[0].tolist(), lts[0].tolist())
    sc = S.build_schedule(times, tts[0].tolist(), dt, True)
    hx = torch.randn(8, h, w, C, device="cuda") * 0.5
    e = torch.randn(sc.n_draws, h, w, C, device="cuda")
    print(f"rollout 10 steps + 8 jumps      : {timeit(lambda: ode.rollout_nhwc(hx, sc, e), 5, 2):9.1f} us")
    if "--bigconvs" in sys.argv:
        for args in [(64, 64, 3, 7, 200, 200), (128, 64, 3, 1, 200, 200, 1, 64), (128, 128, 3, 1, 200, 200, 1, 64), (64, 128, 3, 7, 200, 200, 12),
                     (512, 128, 1, 7, 200, 200), (128, 128, 3, 7, 200, 200), (256, 256, 3, 8, 50, 50), (128, 128, 3, 7, 100, 100), (64, 256, 1, 7, 200, 200)]:
            conv_case(*args)
        return
    if "--nsweep" in sys.argv:
        # batch dependence of the large-tile kernels: 7 frames live in the 256 MB Infinity Cache, 224 stream from HBM
        for n in (7, 28, 112, 224):
            conv_case(128, 128, 3, n, 200, 200)
        for n in (7, 28, 112, 224):
            conv_case(64, 64, 3, n, 200, 200)
        for n in (7, 224):
            conv_case(128, 64, 3, n, 200, 200, 1, 64)
        return
    if "--fixedcost" in sys.argv:
        # time = rounds x (a + c x chunks): the same layer at K = 1, 2, 4, 8, 36 chunks of 32 (1x1 convs of 32..256 channels, 3x3 of 128)
        for cin in (32, 64, 128, 256):
            conv_case(cin, 128, 1, 112, 200, 200)
        conv_case(128, 128, 3, 112, 200, 200)
        for cin in (32, 64, 128, 256):
            conv_case(cin, 64, 1, 112, 200, 200)
        conv_case(64, 64, 3, 112, 200, 200)
        return
    if "--tail" in sys.argv:
        # tail-quantisation probe: 256x256 images = 1024 64-pixel-row tiles each
        for n in (1, 2, 3, 4, 5, 8, 16):
            conv_case(64, 64, 3, n, 256, 256)
        for n in (1, 2, 4, 8, 16):
            conv_case(128, 128, 3, n, 256, 256)
        for hw in (232, 240, 248, 256, 264, 272):
            conv_case(128, 128, 3, 4, hw, hw)
        return
    if "--convs" in sys.argv:
        for args in [(8, 8, 1, 1, 4, 4), (64, 64, 1, 1, 4, 4), (64, 64, 1, 1, 50, 50), (64, 64, 3, 1, 50, 50), (128, 64, 3, 1, 50, 50, 1, 64),
                     (128, 128, 3, 1, 50, 50, 1, 64), (64, 128, 3, 1, 50, 50), (128, 128, 3, 1, 50, 50), (128, 64, 7, 1, 50, 50, 1, 64),
                     (64, 64, 3, 1, 200, 200), (128, 128, 3, 1, 200, 200, 1, 64)]:
            conv_case(*args)
        return
    for B in (2, 4, 8, 16):
        hxb = torch.randn(8, B, h, w, C, device="cuda") * 0.5
        eb = torch.randn(sc.n_draws, B, h, w, C, device="cuda")
        t = timeit(lambda: ode.rollout_nhwc(hxb, sc, eb), 3, 1)
        print(f"rollout batch {B}                 : {t:9.1f} us  = {t / B:8.1f} us / sample")
    if "--quick" in sys.argv:
        return
    obs = torch.randn(8, H, W, C, device="cuda")
    print(f"small_encoder 8 frames          : {timeit(lambda: ode.srvp_encoder.forward_nhwc(obs), 5, 2):9.1f} us  (114 GFLOP)")
    lat = torch.randn(7, h, w, C, device="cuda")
    print(f"small_decoder 7 frames          : {timeit(lambda: ode.srvp_decoder.forward_nhwc(lat), 5, 2):9.1f} us  (173 GFLOP)")
    fr = torch.randn(7, H, W, C, device="cuda")
    print(f"spatial_gru 7 frames            : {timeit(lambda: net.spatial_grus[0].forward_nhwc(fr, fr[0]), 5, 2):9.1f} us  (126 GFLOP)")
    print(f"convnext block 7 frames         : {timeit(lambda: net.res_blocks[0][0].forward_nhwc(fr), 5, 2):9.1f} us  (20 GFLOP)")
    print(f"deeplab head 7 frames           : {timeit(lambda: net.res_blocks[1].forward_nhwc(fr), 5, 2):9.1f} us  (261 GFLOP)")
    xin = torch.randn(8, C, H, W, device="cuda")
    print(f"nchw->nhwc 8 frames             : {timeit(lambda: runtime.to_nhwc(xin), 5, 2):9.1f} us")
    print()
    for args in [(128, 128, 3, 1, 50, 50, 1, 64), (64, 64, 3, 1, 50, 50), (128, 64, 7, 1, 50, 50, 1, 64), (64, 64, 1, 1, 50, 50),
                 (64, 128, 3, 1, 50, 50), (128, 128, 3, 1, 50, 50),
                 (128, 128, 3, 1, 200, 200, 1, 64), (64, 64, 3, 7, 200, 200), (64, 128, 3, 7, 200, 200, 12),
                 (512, 128, 1, 7, 200, 200), (128, 128, 3, 7, 200, 200), (256, 256, 3, 8, 50, 50), (128, 128, 3, 7, 100, 100)]:
        conv_case(*args)


if __name__ == "__main__":
    main()
