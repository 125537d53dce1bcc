"""GPU: randomised parity sweep of the fused convolution entry points (sf_conv2d_ex_fwd) against torch CPU fp32:
random channel splits (two concatenated inputs), kernel sizes, strides, dilations, paddings, batch sizes, odd
spatial sizes, channel-sliced inputs / outputs, residual before or after the activation.  The seeds are fixed:
the same 48 configurations every run; they cover every tile configuration the dispatcher can choose
(direct-fragment, split-K, 64x64, 128x128)."""
import ctypes
import os
import random

import pytest
import torch
import torch.nn.functional as F

from util import hashfill, maxabs

pytestmark = pytest.mark.gpu


def _cfg(i):
    r = random.Random(1000 + i)
    big = i % 6 == 0
    k = r.choice([1, 3, 3, 3, 5, 7])
    stride = r.choice([1, 1, 1, 2])
    dil = r.choice([1, 1, 2, 3]) if k > 1 and stride == 1 else 1
    c0 = r.choice([4, 8, 12, 32, 64, 72])
    c1 = r.choice([0, 0, 8, 32, 64]) if not big else 0
    cout = r.choice([4, 8, 20, 64, 128, 136]) if not big else r.choice([64, 128])
    n = r.choice([1, 2, 3]) if not big else r.choice([4, 8])
    H = r.choice([5, 12, 17, 33, 50]) if not big else r.choice([150, 200])
    W = r.choice([7, 12, 21, 40, 50]) if not big else r.choice([160, 200])
    pad = r.choice([dil * (k - 1) // 2, 0]) if k > 1 else 0
    if (H + 2 * pad - dil * (k - 1) - 1) // stride + 1 < 1 or (W + 2 * pad - dil * (k - 1) - 1) // stride + 1 < 1:
        pad = dil * (k - 1) // 2
    return dict(k=k, stride=stride, dil=dil, c0=c0, c1=c1, cout=cout, n=n, H=H, W=W, pad=pad, act=r.choice(["none", "relu", "lrelu", "tanh"]),
                add=r.random() < 0.6, after=r.random() < 0.5, in_slack=r.choice([0, 8]), out_slack=r.choice([0, 4, 16]))


@pytest.mark.parametrize("i", range(48))
def test_random_conv(i):
    _run(_cfg(i), i)


# Layers the dispatcher sends to the LDS-DMA kernel (conv_glds_kernel: channel counts multiples of 32, >= 640 tiles of
# 64x64, no split-K): both tile families (cout 64 / 128, below and above 131072 pixels), two concatenated sources,
# stride 2, dilation, nearest x2 upsampling on read, no padding, odd sizes whose last tile is ragged, and many small
# images so that one 128-pixel tile spans several of them.
_DMA = [
    dict(k=3, stride=1, dil=1, c0=64, c1=64, cout=64, n=2, H=181, W=187, pad=1),
    dict(k=3, stride=1, dil=1, c0=32, c1=96, cout=128, n=4, H=150, W=231, pad=1),
    dict(k=3, stride=2, dil=1, c0=64, c1=0, cout=128, n=3, H=301, W=403, pad=1),
    dict(k=3, stride=1, dil=12, c0=64, c1=0, cout=128, n=4, H=200, W=200, pad=12),
    dict(k=1, stride=1, dil=1, c0=96, c1=32, cout=64, n=5, H=200, W=173, pad=0),
    dict(k=5, stride=1, dil=1, c0=32, c1=0, cout=192, n=2, H=120, W=131, pad=0),
    dict(k=3, stride=1, dil=1, c0=64, c1=0, cout=64, n=2, H=100, W=100, pad=1, in_up=1),
    dict(k=3, stride=1, dil=1, c0=32, c1=32, cout=128, n=2, H=130, W=140, pad=1, in_up=1),
    dict(k=3, stride=1, dil=1, c0=64, c1=64, cout=128, n=600, H=15, W=17, pad=1),
    dict(k=3, stride=1, dil=2, c0=64, c1=0, cout=64, n=3000, H=7, W=7, pad=2),
    dict(k=7, stride=1, dil=1, c0=64, c1=64, cout=64, n=17, H=50, W=50, pad=3),
    dict(k=3, stride=1, dil=1, c0=128, c1=0, cout=256, n=8, H=50, W=50, pad=1),
]


@pytest.mark.parametrize("i", range(len(_DMA)))
def test_dma_conv(i):
    c = dict(act=["none", "relu", "lrelu", "tanh"][i % 4], add=i % 3 != 0, after=i % 2 == 0, in_slack=8 * (i % 2), out_slack=[0, 4, 16][i % 3])
    c.update(_DMA[i])
    from streamingflow_amd import _lib
    L = _lib.lib()
    NK = _lib.SF_PROF_KEYS
    calls, ms = (ctypes.c_int32 * NK)(), (ctypes.c_double * NK)()
    fl, by = (ctypes.c_double * NK)(), (ctypes.c_double * NK)()
    L.sf_prof_enable(1)
    try:
        _run(c, 100 + i)
        torch.cuda.synchronize()
        L.sf_prof_collect(calls, ms, fl, by)
    finally:
        L.sf_prof_enable(0)
    used = [_lib.KERNEL_NAMES[k] for k in range(NK) if calls[k]]
    assert used and all(u.startswith("conv_glds") for u in used), used   # the case really ran on the LDS-DMA kernel


def _run(c, i, tol=2e-4, wino=False):
    from streamingflow_amd import _lib, packing, runtime
    was = packing.winograd()
    packing.set_winograd(wino)      # the direct-form tests pack without Winograd weights (several of their layers would qualify)
    try:
        return _run_packed(c, i, tol)
    finally:
        packing.set_winograd(was)


def _run_packed(c, i, tol):
    from streamingflow_amd import _lib, packing, runtime
    k, n, H, W, c0, c1, cout = c["k"], c["n"], c["H"], c["W"], c["c0"], c["c1"], c["cout"]
    up = c.get("in_up", 0)
    x0 = hashfill.normal(f"rc_x0_{i}", (n, c0, H, W), 1)
    x1 = hashfill.normal(f"rc_x1_{i}", (n, c1, H, W), 2) if c1 else None
    w = hashfill.uniform(f"rc_w_{i}", (cout, c0 + c1, k, k), -1, 1, 3) * (3.0 / ((c0 + c1) * k * k)) ** 0.5
    b = hashfill.uniform(f"rc_b_{i}", (cout,), -0.5, 0.5, 4)
    sc = hashfill.uniform(f"rc_s_{i}", (cout,), 0.5, 1.5, 5)
    xin = torch.cat([x0, x1], 1) if c1 else x0
    if up:
        xin = F.interpolate(xin, scale_factor=2, mode="nearest")
    y = F.conv2d(xin, w, None, c["stride"], c["pad"], c["dil"]) * sc.view(1, -1, 1, 1) + b.view(1, -1, 1, 1)
    Ho, Wo = y.shape[-2:]
    add = hashfill.normal(f"rc_a_{i}", (n, cout, Ho, Wo), 6) if c["add"] else None
    act = {"none": lambda t: t, "relu": F.relu, "lrelu": lambda t: F.leaky_relu(t, 0.1), "tanh": torch.tanh, "gelu": F.gelu}[c["act"]]
    want = act(y + add) if (c["add"] and c["after"]) else (act(y) + add if c["add"] else act(y))

    pk = packing.Pack(None)
    cw = packing.conv_w(pk, w.cuda(), c0, c1, sc.cuda(), b.cuda(), c["act"], dil=c["dil"], stride=c["stride"], pad=c["pad"])
    # inputs live inside wider NHWC tensors (channel stride > channels), output goes to a channel slice
    cs0 = c0 + c["in_slack"]
    a0 = torch.randn((n, H, W, cs0), device="cuda")
    a0[..., :c0] = x0.permute(0, 2, 3, 1).cuda()
    a1 = None
    if c1:
        a1 = torch.randn((n, H, W, c1 + 4), device="cuda")
        a1[..., :c1] = x1.permute(0, 2, 3, 1).cuda()
    ocs, oco = cout + c["out_slack"], c["out_slack"] // 2 // 4 * 4
    out = torch.full((n, Ho, Wo, ocs), 7.0, device="cuda")
    addn = add.permute(0, 2, 3, 1).contiguous().cuda() if add is not None else None
    L = _lib.lib()
    ws = runtime.workspace(L.sf_conv2d_ex_ws_bytes(), "cuda")
    _lib.check(L.sf_conv2d_ex_fwd(ctypes.byref(cw), runtime.ptr(a0), cs0, runtime.ptr(a1), c1 + 4 if c1 else 0, runtime.ptr(addn), cout,
                                  int(c["after"]), ctypes.c_void_p(out.data_ptr()), ocs, oco, n, H, W, up, runtime.ptr(ws), ws.numel() * 4,
                                  runtime.stream_ptr()), "conv2d_ex")
    got = out[..., oco:oco + cout].permute(0, 3, 1, 2)
    assert maxabs(got, want) <= tol, c
    # nothing outside the output slice was touched
    if ocs > cout:
        rest = torch.cat([out[..., :oco], out[..., oco + cout:]], -1)
        assert float((rest - 7.0).abs().max()) == 0.0
    return got.clone()


# Large plain 1x1 layers (the ASPP projection and its 1x1 branch, classifiers, SpatialGRU decoders): K from 32 to 512, ragged pixel
# counts, inputs inside wider tensors, outputs into channel slices, residual before and after the activation, GELU included.
_PW = [
    dict(c0=64, cout=64, n=2, H=181, W=187, act="none"),
    dict(c0=64, cout=128, n=2, H=200, W=200, act="relu"),
    dict(c0=512, cout=128, n=2, H=200, W=173, act="relu"),
    dict(c0=128, cout=64, n=3, H=150, W=150, act="lrelu"),
    dict(c0=32, cout=64, n=1, H=256, W=257, act="gelu"),
    dict(c0=96, cout=128, n=5, H=120, W=131, act="none"),
]


@pytest.mark.parametrize("i", range(len(_PW)))
def test_large_pointwise_conv(i):
    c = dict(k=1, stride=1, dil=1, c1=0, pad=0, add=i % 3 != 1, after=i % 2 == 0, in_slack=8 * (i % 2), out_slack=[0, 4, 16][i % 3])
    c.update(_PW[i])
    _run(c, 500 + i, tol=5e-5)


@pytest.mark.parametrize("i", [0, 1, 3, 8])
def test_dma_conv_is_bitwise_reproducible(i):
    """A race between the LDS-DMA writes, the fragment reads and the mid-stream barrier would show up as run-to-run
    differences: 20 launches of the same layer (other kernels interleaved to vary the timing) must agree bit for bit."""
    from streamingflow_amd import _lib, packing, runtime
    c = _DMA[i]
    n, H, W, c0, c1, cout, k = c["n"], c["H"], c["W"], c["c0"], c["c1"], c["cout"], c["k"]
    g = torch.Generator(device="cuda").manual_seed(1234 + i)
    a0 = torch.randn((n, H, W, c0), device="cuda", generator=g)
    a1 = torch.randn((n, H, W, c1), device="cuda", generator=g) if c1 else None
    w = torch.randn((cout, c0 + c1, k, k), device="cuda", generator=g) * 0.05
    pk = packing.Pack(None)
    cw = packing.conv_w(pk, w, c0, c1, act="lrelu", dil=c["dil"], stride=c["stride"], pad=c["pad"])
    Ho = (H + 2 * c["pad"] - c["dil"] * (k - 1) - 1) // c["stride"] + 1
    Wo = (W + 2 * c["pad"] - c["dil"] * (k - 1) - 1) // c["stride"] + 1
    L = _lib.lib()
    ws = runtime.workspace(L.sf_conv2d_ex_ws_bytes(), "cuda")
    outs = []
    for r in range(20):
        out = torch.empty((n, Ho, Wo, cout), device="cuda")
        _lib.check(L.sf_conv2d_ex_fwd(ctypes.byref(cw), runtime.ptr(a0), c0, runtime.ptr(a1), c1, None, cout, 0,
                                      ctypes.c_void_p(out.data_ptr()), cout, 0, n, H, W, 0, runtime.ptr(ws), ws.numel() * 4,
                                      runtime.stream_ptr()), "conv2d_ex")
        if r % 3 == 1:
            torch.randn(1 << (10 + r), device="cuda").sum()      # unrelated work in between
        outs.append(out)
    torch.cuda.synchronize()
    for o in outs[1:]:
        assert torch.equal(o, outs[0])


# Layers that qualify for the Winograd F(2x2, 3x3) kernel (csrc/conv_wino.hip: 3x3, stride 1, pad = dilation, inputs in whole
# 16-channel chunks, cout a multiple of 64, >= 14000 pixels): one and two sources, odd sizes (ragged tile blocks at the right / bottom
# edge, odd H / W: a half-used last Winograd tile), two cout tiles, many small images, channel-sliced inputs / outputs, every
# activation, the residual before and after it.  Same oracle and tolerance as the direct form.
_WINO = [
    dict(c0=128, c1=0, cout=128, n=4, H=200, W=200),
    dict(c0=64, c1=64, cout=128, n=5, H=181, W=187),
    dict(c0=64, c1=0, cout=128, n=9, H=100, W=151),
    dict(c0=32, c1=96, cout=256, n=4, H=150, W=231),
    dict(c0=16, c1=48, cout=128, n=100, H=40, W=41),
    dict(c0=256, c1=0, cout=128, n=60, H=50, W=50),
    # cout not a multiple of 128: 64 cout x 64 Winograd tiles per workgroup (one V buffer, transform between the chunks)
    dict(c0=64, c1=0, cout=64, n=4, H=200, W=200),
    dict(c0=64, c1=64, cout=64, n=5, H=181, W=187),
    dict(c0=128, c1=0, cout=192, n=4, H=150, W=231),
    dict(c0=32, c1=0, cout=64, n=30, H=67, W=70),
    # dilated (pad = dilation: the ASPP branches): d x d interleaved phases, tile lists per axis, blocks spanning several phases;
    # phases of two lengths, odd phase lengths (half-used last tile), the largest dilation the kernel takes (5 pixels per phase)
    dict(c0=64, c1=0, cout=128, n=4, H=200, W=200, dil=12, pad=12),
    dict(c0=64, c1=0, cout=128, n=4, H=200, W=200, dil=24, pad=24),
    dict(c0=64, c1=0, cout=128, n=4, H=200, W=200, dil=36, pad=36),
    dict(c0=32, c1=0, cout=64, n=6, H=157, W=171, dil=7, pad=7),
    dict(c0=32, c1=0, cout=192, n=3, H=203, W=241, dil=40, pad=40),
    dict(c0=64, c1=0, cout=64, n=4, H=181, W=187, dil=2, pad=2),
    dict(c0=64, c1=0, cout=64, n=44, H=50, W=64, dil=3, pad=3),
    # nearest x2 upsampling on read (the decoder's upsampling layers): H, W are the INPUT size here
    dict(c0=128, c1=0, cout=128, n=3, H=100, W=100, in_up=1),
    dict(c0=32, c1=32, cout=64, n=4, H=67, W=71, in_up=1),
    # a single frame with 64 output channels: fewer 32-tile workgroups than the chip has slots -> the 16-tile blocks (round 6; 4 x 16
    # output pixels per workgroup: ragged bottom rows of two tile rows instead of four)
    dict(c0=64, c1=0, cout=64, n=1, H=200, W=200),
    dict(c0=64, c1=64, cout=64, n=1, H=181, W=187),
    dict(c0=32, c1=0, cout=64, n=2, H=150, W=131),
]


@pytest.mark.parametrize("i", range(len(_WINO)))
def test_winograd_conv(i):
    if os.environ.get("SF_WINO") == "0":
        pytest.skip("SF_WINO=0: the library keeps the direct form everywhere")
    c = dict(k=3, stride=1, dil=1, pad=1, act=["relu", "none", "lrelu", "tanh"][i % 4], add=i % 3 != 1, after=i % 2 == 0, in_slack=8 * (i % 2),
             out_slack=[0, 4, 16][i % 3])
    c.update(_WINO[i])
    from streamingflow_amd import _lib
    L = _lib.lib()
    NK = _lib.SF_PROF_KEYS
    calls, ms = (ctypes.c_int32 * NK)(), (ctypes.c_double * NK)()
    fl, by = (ctypes.c_double * NK)(), (ctypes.c_double * NK)()
    L.sf_prof_enable(1)
    try:
        got = _run(c, 300 + i, wino=True)
        torch.cuda.synchronize()
        L.sf_prof_collect(calls, ms, fl, by)
    finally:
        L.sf_prof_enable(0)
    used = [_lib.KERNEL_NAMES[k] for k in range(NK) if calls[k]]
    assert used and all(u.startswith("conv_wino") for u in used), used      # the case really ran on the Winograd kernel
    # ... and agrees with the direct form of the same layer far inside the tolerance both have against torch
    direct = _run(c, 300 + i, wino=False)
    assert maxabs(got, direct) <= 5e-5, maxabs(got, direct)


@pytest.mark.parametrize("i", [0, 1, 4, 5, 7, 11, 13])
def test_winograd_conv_is_bitwise_reproducible(i):
    """Two workgroups per CU, LDS-DMA rings with counted waits, transform buffers reused chunk after chunk: a race would show up as
    run-to-run differences.  12 launches of the same layer (other work interleaved) must agree bit for bit — plain, dilated, and narrow
    images concatenated along x (cases 4 and 5)."""
    from streamingflow_amd import _lib, packing, runtime
    c = dict(dil=1)
    c.update(_WINO[i])
    n, H, W, c0, c1, cout, dil = c["n"], c["H"], c["W"], c["c0"], c["c1"], c["cout"], c["dil"]
    g = torch.Generator(device="cuda").manual_seed(4321 + i)
    a0 = torch.randn((n, H, W, c0), device="cuda", generator=g)
    a1 = torch.randn((n, H, W, c1), device="cuda", generator=g) if c1 else None
    w = torch.randn((cout, c0 + c1, 3, 3), device="cuda", generator=g) * 0.05
    was = packing.winograd()
    packing.set_winograd(True)
    try:
        pk = packing.Pack(None)
        cw = packing.conv_w(pk, w, c0, c1, act="lrelu", dil=dil, stride=1, pad=dil)
    finally:
        packing.set_winograd(was)
    assert cw.w_wino, "the layer must have been packed with Winograd weights"
    L = _lib.lib()
    ws = runtime.workspace(L.sf_conv2d_ex_ws_bytes(), "cuda")
    outs = []
    for r in range(12):
        out = torch.empty((n, H, W, cout), device="cuda")
        _lib.check(L.sf_conv2d_ex_fwd(ctypes.byref(cw), runtime.ptr(a0), c0, runtime.ptr(a1), c1, None, cout, 0,
                                      ctypes.c_void_p(out.data_ptr()), cout, 0, n, H, W, 0, runtime.ptr(ws), ws.numel() * 4,
                                      runtime.stream_ptr()), "conv2d_ex")
        if r % 3 == 1:
            torch.randn(1 << (12 + r), device="cuda").sum()      # unrelated work in between
        outs.append(out)
    torch.cuda.synchronize()
    for o in outs[1:]:
        assert torch.equal(o, outs[0])



def _wino_calls(fn):
    """run fn() under the library profiler: (result, names of the kernels that ran)"""
    from streamingflow_amd import _lib
    L = _lib.lib()
    NK = _lib.SF_PROF_KEYS
    calls, ms = (ctypes.c_int32 * NK)(), (ctypes.c_double * NK)()
    fl, by = (ctypes.c_double * NK)(), (ctypes.c_double * NK)()
    L.sf_prof_enable(1)
    try:
        r = fn()
        torch.cuda.synchronize()
        L.sf_prof_collect(calls, ms, fl, by)
    finally:
        L.sf_prof_enable(0)
    return r, {_lib.KERNEL_NAMES[k]: calls[k] for k in range(NK) if calls[k]}


def test_winograd_epilogues_of_the_batched_latents_against_the_direct_form():
    """the epilogue variants the unit cases above do not reach, at kernel level on the launches of 8 batched 50x50 latents
    (images concatenated along x): the conv-GRU cell = [update ; reset] gates with the reset-gate second output (AFFINE + out2 /
    gate_from), then the candidate with the state blend (BLEND); and infer_state = SE-scaled input (the plain form: per-image scales
    staged in LDS), residual with an SE scale (add_scale), the clamped last layer.  Each module runs twice — packed with and without
    Winograd weights — on the same inputs: the profiler must show the Winograd kernel in the first run only, and the two results must
    agree far inside the 1e-3 end-to-end tolerance that would otherwise be the only check of a masked-lane or offset bug."""
    from util import build_pair, cases
    from streamingflow_amd import packing
    C, B, h, w = 64, 8, 50, 50
    cts, lts, tts, dt = cases.timeset("shipped")

    def build(wino):
        was = packing.winograd()
        packing.set_winograd(wino)
        try:
            net, _ = build_pair(C, "euler", True, True, dt)
            gru, ode = net.spatial_grus[0], net.gru_ode
            x = (hashfill.normal("wl_x", (3, B, h, w, C), 11) * 0.5).cuda()          # [T, B, h, w, C] NHWC frames
            s0 = (hashfill.normal("wl_s", (B, h, w, C), 12) * 0.5).cuda()
            ode.noise = hashfill.HashedNoise(3)

            def run():
                y = gru.forward_nhwc(x, s0)
                p, q = ode.infer_state(s0.permute(0, 3, 1, 2).contiguous())
                return y, p, q
            return _wino_calls(run)
        finally:
            packing.set_winograd(was)
    (ya, pa, qa), used_a = build(True)
    (yb, pb, qb), used_b = build(False)
    assert any(k.startswith("conv_wino") and k.endswith("blend>") for k in used_a), used_a
    assert any(k.startswith("conv_wino") and k.endswith("affine>") for k in used_a), used_a
    if os.environ.get("SF_WINO_SAMPLE", "1") != "0":      # round 6: the sampling layer (SE-scaled input, (loc, raw) interleaved rows, eps from the hashed stream)
        assert any(k.startswith("conv_wino") and k.endswith("sample>") for k in used_a), used_a
    assert not any(k.startswith("conv_wino") for k in used_b), used_b
    for a, b in ((ya, yb), (pa, pb), (qa, qb)):
        assert a.shape == b.shape and maxabs(a, b) <= 2e-5, maxabs(a, b)


@pytest.mark.parametrize("thr", [0, 1000000000])
def test_winograd_cases_under_both_block_sizes(thr):
    """conv_wino5_kernel has two block sizes (32 and 16 Winograd tiles per workgroup; launch_conv_wino picks by the launch's workgroup
    count).  Every Winograd case of this file and of test_gpu_ops.py again in a child process with the choice forced: never / always the
    16-tile form — the same oracle, the same tolerances."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env["SF_WINO_SMALL_WGS"] = str(thr)
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-m", "gpu", "-p", "no:cacheprovider", "-x",
                        os.path.join(root, "tests", "test_gpu_conv_random.py"), os.path.join(root, "tests", "test_gpu_ops.py"),
                        "-k", "winograd and not both_block_sizes"], env=env, capture_output=True, text=True, timeout=2400, cwd=root)
    tail = r.stdout[-1500:] + r.stderr[-1500:]
    assert r.returncode == 0 and " passed" in r.stdout and "failed" not in r.stdout, tail
