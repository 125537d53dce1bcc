"""GPU parity, op by op: libsfnative (through the product modules / the C ABI) against
(a) the fixtures generated from the real reference (tests/golden/ops_c8.npz) and
(b) the oracle on the same hashed inputs at the shipped channel count (C=64, 50x50 latent).
Tolerance: 1e-4 max-abs per op, fp32 everywhere."""
import ctypes
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from util import cases, hashfill, gold, maxabs, build_pair
from oracle import ref_torch as R

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def pair8():
    return build_pair(8)


@pytest.fixture(scope="module")
def pair64():
    return build_pair(64)


def _noise():
    return hashfill.HashedNoise(cases.EPS_SEED)


# ---- generic conv through the C ABI ------------------------------------------------------------
@pytest.mark.parametrize("cin,cout,k,dil,n,H,W", [
    (8, 8, 3, 1, 1, 12, 12), (64, 64, 3, 1, 1, 50, 50), (128, 64, 7, 1, 1, 50, 50), (64, 128, 1, 1, 1, 50, 50),
    (64, 128, 3, 12, 2, 96, 100), (16, 24, 3, 1, 3, 33, 47), (64, 64, 3, 36, 1, 200, 200), (40, 72, 3, 2, 1, 20, 21)])
def test_conv2d(cin, cout, k, dil, n, H, W):
    from streamingflow_amd import _lib, packing, runtime
    x = hashfill.normal("cx", (n, cin, H, W), 5).cuda()
    w = hashfill.uniform("cw", (cout, cin, k, k), -1, 1, 6) * (3.0 / (cin * k * k)) ** 0.5
    b = hashfill.uniform("cb", (cout,), -0.5, 0.5, 7)
    add = hashfill.normal("cadd", (n, cout, H, W), 8)
    pk = packing.Pack(None)
    cw = packing.conv_w(pk, w.cuda(), cin, bias=b.cuda(), act="lrelu", dil=dil)
    xn, an = runtime.to_nhwc(x), runtime.to_nhwc(add.cuda())
    out = torch.empty((n, H, W, cout), device="cuda")
    _lib.check(_lib.lib().sf_conv2d_fwd(ctypes.byref(cw), runtime.ptr(xn), None, runtime.ptr(an), runtime.ptr(out),
                                        n, H, W, 0, runtime.stream_ptr()))
    ref = F.leaky_relu(F.conv2d(x.cpu(), w, b, padding=dil * (k - 1) // 2, dilation=dil), 0.1) + add
    assert maxabs(runtime.to_nchw(out), ref) <= TOL


def test_layout_roundtrip():
    from streamingflow_amd import runtime
    x = hashfill.normal("lay", (3, 24, 17, 29), 1).cuda()
    y = runtime.to_nhwc(x)
    assert torch.equal(y, x.permute(0, 2, 3, 1).contiguous())
    assert torch.equal(runtime.to_nchw(y), x)


# ---- golden vectors from the reference (C=8, 12x12 latent / 48x48 BEV) ---------------------------
def test_golden_cells(pair8):
    net, _ = pair8
    g = gold("ops_c8.npz")
    C, h, w = 8, 12, 12
    x = hashfill.normal("op_x", (1, C, h, w), 11).cuda()
    s = (hashfill.normal("op_s", (1, C, h, w), 12) * 0.5).cuda()
    ode = net.gru_ode
    assert maxabs(net.spatial_grus[0].gru_cell(x, s), g["spatial_gru_cell"]) <= TOL
    assert maxabs(ode.gru_c(x, s), g["dual_ode_cell"]) <= TOL
    assert maxabs(ode.gru_obs(s, None, x)[0], g["dual_obs_cell"]) <= TOL
    ode.noise = _noise()
    y, q = ode.infer_state(s)
    assert maxabs(y, g["infer_state_y"]) <= TOL and maxabs(q, g["infer_state_q"]) <= TOL


def test_golden_encoder_decoder(pair8):
    net, _ = pair8
    g = gold("ops_c8.npz")
    C, h, w = 8, 12, 12
    bev = hashfill.normal("op_bev", (1, 2, C, 4 * h, 4 * w), 13).cuda()
    assert maxabs(net.gru_ode.srvp_encode(bev)[0], g["srvp_encode"]) <= TOL
    lat = (hashfill.normal("op_lat", (1, 2, C, h, w), 14) * 0.5).cuda()
    assert maxabs(net.gru_ode.srvp_decode(lat), g["srvp_decode"]) <= TOL


def test_golden_head_blocks(pair8):
    net, _ = pair8
    g = gold("ops_c8.npz")
    C, h, w = 8, 12, 12
    frames = hashfill.normal("op_frames", (3, C, 4 * h, 4 * w), 15).cuda()
    assert maxabs(net.res_blocks[0][0](frames), g["convnext_block"]) <= TOL
    assert maxabs(net.res_blocks[1](frames), g["deeplab_head"]) <= TOL
    seq = hashfill.normal("op_seq", (1, 3, C, 4 * h, 4 * w), 16).cuda()
    assert maxabs(net.spatial_grus[1](seq, seq[:, 0]), g["spatial_gru_seq"]) <= TOL


@pytest.mark.parametrize("n,H,W", [(2, 181, 187), (1, 256, 260), (1, 64, 64), (2, 200, 200), (1, 255, 300)])
def test_convnext_block_c64_vs_torch(n, H, W):
    """C == 64 maps of >= 65536 pixels take a register-window depthwise + LayerNorm kernel — two channels per lane and two strips
    per wave where the strips of an image come in pairs ((1, 256, 260), (2, 200, 200), and (1, 255, 300): odd height, ragged last
    segment), one channel per lane otherwise ((2, 181, 187)) — smaller ones the LDS-tile kernel; all against the reference
    formulation (convolutions.py:310-346) in torch fp32 on the CPU, with gamma = 1 so that the block's body is not scaled away."""
    import streamingflow_amd.layers.convolutions as Cv
    torch.manual_seed(H * 7 + W)
    blk = Cv.Block(64, layer_scale_init_value=1.0).eval()
    with torch.no_grad():
        blk.norm.weight.uniform_(0.5, 1.5); blk.norm.bias.uniform_(-0.5, 0.5)
    x = torch.randn(n, 64, H, W)
    with torch.no_grad():
        y = F.conv2d(x, blk.dwconv.weight, blk.dwconv.bias, padding=3, groups=64).permute(0, 2, 3, 1)
        y = F.layer_norm(y, (64,), blk.norm.weight, blk.norm.bias, 1e-6)
        y = blk.pwconv2(F.gelu(blk.pwconv1(y)))
        ref = x + (blk.gamma * y).permute(0, 3, 1, 2)
        got = blk.cuda()(x.cuda())
    assert maxabs(got, ref) <= TOL


def test_deeplab_head_c64_vs_torch():
    """DeepLabHead(64, 64, 128) on 2 x 200 x 173 frames against the reference formulation (convolutions.py:217-280) in torch fp32 on
    the CPU at the shipped width: the projection takes the pooled branch as a per-IMAGE bias (the two frames get different rows), the
    dilated branches and the 3x3 run the Winograd kernel."""
    import ctypes
    import streamingflow_amd.layers.convolutions as Cv
    from streamingflow_amd import _lib
    torch.manual_seed(77)
    head = Cv.DeepLabHead(64, 64, 128).eval()
    with torch.no_grad():
        for m in head.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.uniform_(0.7, 1.3); m.bias.uniform_(-0.2, 0.2)
                m.running_mean.uniform_(-0.2, 0.2); m.running_var.uniform_(0.5, 1.5)
    x = torch.randn(2, 64, 200, 173)
    x[1] += 0.7                      # different pooled vectors per frame
    with torch.no_grad():
        aspp = head[0]
        outs = [b(x) for b in list(aspp.convs)[:4]]
        outs.append(F.interpolate(aspp.convs[4](x), size=x.shape[-2:], mode="bilinear", align_corners=False))
        ref = head[4](head[3](head[2](head[1](aspp.project(torch.cat(outs, 1))))))
        head = head.cuda()
        head(x.cuda())               # packs
        L = _lib.lib()
        NK = _lib.SF_PROF_KEYS
        calls, ms = (ctypes.c_int32 * NK)(), (ctypes.c_double * NK)()
        fl, by = (ctypes.c_double * NK)(), (ctypes.c_double * NK)()
        L.sf_prof_enable(1)
        try:
            got = head(x.cuda())
            torch.cuda.synchronize()
            L.sf_prof_collect(calls, ms, fl, by)
        finally:
            L.sf_prof_enable(0)
    used = {_lib.KERNEL_NAMES[k]: calls[k] for k in range(NK) if calls[k]}
    assert used.get("conv_wino<64x32t2dil,affine>", 0) >= 2 and used.get("conv_wino<64x32t2,affine>", 0) >= 1, used
    assert maxabs(got, ref) <= 2e-4, maxabs(got, ref)


@pytest.mark.parametrize("n,H,W", [(2, 104, 120), (3, 96, 64), (2, 102, 118)])
def test_encoder_pools_in_the_winograd_epilogue(pair64, n, H, W):
    """SmallEncoder (res_models.py:98-109) at the shipped width on frames large enough for the Winograd kernel: the MaxPool2d(2)
    after blocks 0 and 1 is taken over the epilogue's own 2x2 tile (no full-size tensor, no pooling launch) — asserted through the
    profiler (no direct-form 3x3 launch) and against the oracle's encoder with the same weights.  (2, 102, 118): 51 x 59 after the
    first pool, odd: the second pool falls back to the pooling launch (floor mode)."""
    import ctypes
    from streamingflow_amd import _lib
    net, sd = pair64
    x = hashfill.normal("enc_pool_x", (n, 64, H, W), 31)
    want = R.small_encoder(sd, "gru_ode.srvp_encoder", x)
    L = _lib.lib()
    NK = _lib.SF_PROF_KEYS
    calls, ms = (ctypes.c_int32 * NK)(), (ctypes.c_double * NK)()
    fl, by = (ctypes.c_double * NK)(), (ctypes.c_double * NK)()
    with torch.no_grad():
        net.gru_ode.srvp_encoder(x.cuda())          # packs
        L.sf_prof_enable(1)
        try:
            got = net.gru_ode.srvp_encoder(x.cuda())
            torch.cuda.synchronize()
            L.sf_prof_collect(calls, ms, fl, by)
        finally:
            L.sf_prof_enable(0)
    used = {_lib.KERNEL_NAMES[k]: calls[k] for k in range(NK) if calls[k]}
    assert any(k.startswith("conv_wino") for k in used), used
    assert maxabs(got, want) <= 2e-4, maxabs(got, want)


@pytest.mark.parametrize("n,h,w", [(2, 26, 30), (5, 25, 25), (1, 50, 50)])
def test_decoder_reads_the_upsampled_skip_in_the_winograd_epilogue(pair64, n, h, w):
    """SmallDecoder (res_models.py:131-145) at the shipped width on latents whose 4x output is large enough for the Winograd kernel:
    the last residual block reads its input upsampled on the fly — the first convolution in its patch loads, the identity skip in
    the second convolution's epilogue (one half-size source pixel per 2x2 tile) — so no upsampled copy is written.  Against the
    oracle's decoder with the same weights."""
    net, sd = pair64
    z = hashfill.normal("dec_up_z", (n, 64, h, w), 32) * 0.5
    want = R.small_decoder(sd, "gru_ode.srvp_decoder", z)
    with torch.no_grad():
        got = net.gru_ode.srvp_decoder(z.cuda())
    assert maxabs(got, want) <= 2e-4, maxabs(got, want)


@pytest.mark.parametrize("which", ["pair8", "pair64"])
def test_head_with_the_last_decoder_folded_into_the_aspp(which, request, monkeypatch):
    """FuturePredictionODE.head_nhwc (future_prediction_ode.py:56-62) hands the hidden states of the last SpatialGRU to a
    DeepLabHead whose branch and pooling weights carry conv_decoder (bias-free 1x1: linear, maps the zero padding to zeros), so the
    decoder launches never run.  Against the same head with the decoder run per frame: equal to fp32 rounding of the composed
    weights, far inside the 1e-3 bar; the planar-output form of both too."""
    import streamingflow_amd.models.future_prediction_ode as M
    net, _ = request.getfixturevalue(which)
    C = 8 if which == "pair8" else 64
    T, B, H, W = 3, 2, 48, 40
    x = hashfill.normal("head_fold_x", (T, B, H, W, C), 21).cuda()
    outs = {}
    for fold in (False, True):
        monkeypatch.setattr(M, "_FOLD_DECODER", fold)
        with torch.no_grad():
            y = net.head_nhwc(x)
            res = torch.empty((B, T, C, H, W), device="cuda")
            assert net.head_nhwc(x, res) is None
        outs[fold] = (y, res)
    scale = float(outs[False][0].abs().max())
    assert maxabs(outs[True][0], outs[False][0]) <= 2e-5 * max(1.0, scale)
    assert maxabs(outs[True][1], outs[False][1]) <= 2e-5 * max(1.0, scale)
    assert maxabs(outs[True][1], outs[True][0].permute(1, 0, 4, 2, 3)) <= 1e-5 * max(1.0, scale)


@pytest.mark.parametrize("T,B,H,W", [(3, 2, 64, 72), (2, 3, 33, 47), (1, 1, 200, 173)])
def test_deeplab_head_writes_the_boundary_layout_itself(T, B, H, W):
    """sf_deeplab_head_planar_fwd: frames (t, b) of a [T][B] run land in a [B, T, C, H, W] tensor as the reference returns it
    (future_prediction_ode.py:62-64), written by the classifier's own epilogue.  Same arithmetic as the [pixel][channel] form followed
    by a transpose, so the two must agree to the last bit when the same kernel runs both (and to rounding when the small-P kernel,
    which has no planar store, ran the [pixel][channel] form); nothing outside the frames may be touched."""
    import streamingflow_amd.layers.convolutions as Cv
    from streamingflow_amd import runtime
    torch.manual_seed(T * 100 + B)
    head = Cv.DeepLabHead(64, 64, 128).eval().cuda()
    x = torch.randn(T * B, H, W, 64, device="cuda")
    with torch.no_grad():
        want = runtime.to_nchw(head.forward_nhwc(x)).view(T, B, 64, H, W).permute(1, 0, 2, 3, 4)
        res = torch.full((B, T + 1, 64, H, W), 7.0, device="cuda")            # one spare frame per sample: must stay untouched
        head.forward_nhwc_into_planar(x, res, B, 64 * H * W, (T + 1) * 64 * H * W)
    if T * B * H * W >= 12288:
        assert torch.equal(res[:, :T], want)
    else:
        assert maxabs(res[:, :T], want) <= 1e-5
    assert float((res[:, T] - 7.0).abs().max()) == 0.0


@pytest.mark.parametrize("n,H,W", [(1, 5, 5), (1, 8, 8), (3, 37, 41), (2, 100, 100)])
def test_convnext_mlp_in_one_launch(n, H, W):
    """pwconv1 -> GELU -> pwconv2 -> gamma -> residual of a 64-channel block runs as ONE launch (convnext_mlp.hip: the 256-channel
    hidden tensor stays in registers): the profiler must see that kernel and no other GEMM launch, and the block must agree with the
    reference formulation (convolutions.py:333-346) in torch fp64 far inside the 1e-3 bar — sizes below one 64-pixel wave tile, an
    exact multiple of it, and ragged ones; random gamma, LayerNorm affine and biases so that no term is scaled away."""
    import ctypes
    import streamingflow_amd.layers.convolutions as Cv
    from streamingflow_amd import _lib
    torch.manual_seed(1000 * n + H)
    blk = Cv.Block(64, layer_scale_init_value=1.0).eval()
    with torch.no_grad():
        blk.norm.weight.uniform_(0.5, 1.5); blk.norm.bias.uniform_(-0.5, 0.5)
        blk.gamma.uniform_(-1.5, 1.5); blk.pwconv1.bias.uniform_(-1.0, 1.0); blk.pwconv2.bias.uniform_(-1.0, 1.0)
        blk.pwconv1.weight.mul_(3.0)          # hidden values out to |v| ~ 6: both tails of the GELU
    x = torch.randn(n, 64, H, W)
    with torch.no_grad():
        d = blk.double()
        xd = x.double()
        y = F.conv2d(xd, d.dwconv.weight, d.dwconv.bias, padding=3, groups=64).permute(0, 2, 3, 1)
        y = F.layer_norm(y, (64,), d.norm.weight, d.norm.bias, 1e-6)
        y = d.pwconv2(F.gelu(d.pwconv1(y)))
        ref = (xd + (d.gamma * y).permute(0, 3, 1, 2)).float()
        blk = blk.float().cuda()
        L = _lib.lib()
        NK = _lib.SF_PROF_KEYS
        calls, ms = (ctypes.c_int32 * NK)(), (ctypes.c_double * NK)()
        fl, by = (ctypes.c_double * NK)(), (ctypes.c_double * NK)()
        blk(x.cuda())                          # packs
        L.sf_prof_enable(1)
        try:
            got = blk(x.cuda())
            torch.cuda.synchronize()
            L.sf_prof_collect(calls, ms, fl, by)
        finally:
            L.sf_prof_enable(0)
    used = [_lib.KERNEL_NAMES[k] for k in range(NK) if calls[k]]
    assert used == ["convnext_mlp<64-256-64,gelu+residual>"], used
    assert maxabs(got, ref) <= 2e-5, maxabs(got, ref)


@pytest.mark.parametrize("solver", ["euler", "midpoint"])
@pytest.mark.parametrize("impute", [True, False])
def test_golden_ode_step(pair8, solver, impute):
    net, _ = pair8
    g = gold("ops_c8.npz")
    C, h, w = 8, 12, 12
    x = hashfill.normal("op_x", (1, C, h, w), 11).cuda()
    s = (hashfill.normal("op_s", (1, C, h, w), 12) * 0.5).cuda()
    ode = net.gru_ode
    old = ode.solver, ode.impute
    try:
        ode.solver, ode.impute = solver, impute
        for dt in (0.05, torch.tensor(0.37, dtype=torch.float64)):
            ode.noise = _noise()
            st, inp, ct, _, _ = ode.ode_step(s, x, dt, 0.0)
            tag = f"ode_step_{solver}_{'imp' if impute else 'noimp'}_{float(dt):.2f}"
            assert maxabs(st, g[tag + "_state"]) <= TOL, tag
            assert maxabs(inp, g[tag + "_input"]) <= TOL, tag
            assert float(ct) == float(dt)
    finally:
        ode.solver, ode.impute = old
        ode.noise = None


# ---- oracle at the shipped size: C=64, latent 50x50 ---------------------------------------------
def test_c64_cells_vs_oracle(pair64):
    net, sd = pair64
    C, h, w = 64, 50, 50
    x = hashfill.normal("x64", (1, C, h, w), 21)
    s = hashfill.normal("s64", (1, C, h, w), 22) * 0.5
    ode = net.gru_ode
    with torch.no_grad():
        assert maxabs(ode.gru_c(x.cuda(), s.cuda()), R.dual_cell(sd, "gru_ode.gru_c", x, s, True)) <= TOL
        assert maxabs(ode.gru_obs(s.cuda(), None, x.cuda())[0], R.dual_cell(sd, "gru_ode.gru_obs.gru_d", x, s, False)) <= TOL
        ode.noise = _noise()
        y, q = ode.infer_state(s.cuda())
        yr, qr = R.infer_state(sd, "gru_ode", s, _noise())
        assert maxabs(q, qr) <= TOL and maxabs(y, yr) <= TOL
        for solver in ("euler", "midpoint", "rk4"):
            ode.solver, ode.noise = solver, _noise()
            st, inp, *_ = ode.ode_step(s.cuda(), x.cuda(), 0.5, 0.0)
            sr, ir = R.ode_step(sd, "gru_ode", s, x, 0.5, solver, True, _noise())
            assert maxabs(st, sr) <= TOL and maxabs(inp, ir) <= TOL, solver
        ode.solver, ode.noise = "euler", None


_TAP_SMALL = [(8, 50, 50), (9, 37, 45), (1, 200, 200), (3, 101, 75)]      # below the form's default size: reached with SF_WINO_LN7_MIN_P=0


@pytest.mark.parametrize("B,h,w", [(32, 50, 50), (27, 37, 67), (1, 262, 258)] + _TAP_SMALL)
def test_batched_cells_7x7_in_winograd_tap_groups_vs_oracle(pair64, B, h, w):
    """Round 6: with >= 65536 pixels per launch the trusting gate's 7x7 + LayerNorm + GELU layer runs on conv_wino5_kernel as nine 3x3 tap
    groups (csrc/conv_wino.hip, GRP = 9; images concatenated along x when 8 tile columns fit them badly).  Both dual cells on batched
    latents — 32 x 50x50 (the headline's launch), odd sizes (ragged tiles, seams between images under every shift of a tap group), one
    large image — against the oracle, and the profiler must show the launch.  The small cases run on the direct form here and on the tap
    groups in test_tap_groups_on_small_launches (a child process with the size rule off)."""
    from streamingflow_amd import _lib
    net, sd = pair64
    C = 64
    x = hashfill.normal(f"x7g{B}", (B, C, h, w), 41)
    s = hashfill.normal(f"s7g{B}", (B, C, h, w), 42) * 0.5
    ode = net.gru_ode
    L = _lib.lib()
    NK = _lib.SF_PROF_KEYS
    calls, ms, fl, by = (ctypes.c_int32 * NK)(), (ctypes.c_double * NK)(), (ctypes.c_double * NK)(), (ctypes.c_double * NK)()
    L.sf_prof_enable(1)
    try:
        with torch.no_grad():
            a = ode.gru_c(x.cuda(), s.cuda())
            b = ode.gru_obs(s.cuda(), None, x.cuda())[0]
        torch.cuda.synchronize()
        L.sf_prof_collect(calls, ms, fl, by)
    finally:
        L.sf_prof_enable(0)
    used = {_lib.KERNEL_NAMES[k]: calls[k] for k in range(NK) if calls[k]}
    if os.environ.get("SF_WINO_LN7", "1") != "0" and os.environ.get("SF_WINO", "1") != "0":
        big = B * h * w >= float(os.environ.get("SF_WINO_LN7_MIN_P", "65536"))
        assert used.get("conv_wino<64x32t2,ln_gelu>", 0) == (2 if big else 0), used
    with torch.no_grad():
        for i in range(B):
            assert maxabs(a[i:i + 1], R.dual_cell(sd, "gru_ode.gru_c", x[i:i + 1], s[i:i + 1], True)) <= TOL, i
            assert maxabs(b[i:i + 1], R.dual_cell(sd, "gru_ode.gru_obs.gru_d", x[i:i + 1], s[i:i + 1], False)) <= TOL, i


def test_tap_groups_on_small_launches():
    """The small cases of the test above with SF_WINO_LN7_MIN_P=0: every one of them on the tap-group form (the assertion on the profiler's
    kernel list follows the variable)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env["SF_WINO_LN7_MIN_P"] = "0"
    ids = " or ".join(f"{B}-{h}-{w}" for B, h, w in _TAP_SMALL)
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-m", "gpu", "-p", "no:cacheprovider", "-x", os.path.join(root, "tests", "test_gpu_ops.py"),
                        "-k", f"tap_groups_vs_oracle and ({ids})"], env=env, capture_output=True, text=True, timeout=1800, cwd=root)
    tail = r.stdout[-1500:] + r.stderr[-1500:]
    assert r.returncode == 0 and f"{len(_TAP_SMALL)} passed" in r.stdout, tail


def test_c64_stress_latent_200(pair64):
    """ode_step fed a 200x200x64 latent directly (SURVEY §8d stress variant; large-tile kernels)."""
    net, sd = pair64
    C, h, w = 64, 200, 200
    x = hashfill.normal("x64b", (1, C, h, w), 23)
    s = hashfill.normal("s64b", (1, C, h, w), 24) * 0.5
    ode = net.gru_ode
    ode.noise = _noise()
    st, inp, *_ = ode.ode_step(s.cuda(), x.cuda(), 0.2, 0.0)
    ode.noise = None
    with torch.no_grad():
        sr, ir = R.ode_step(sd, "gru_ode", s, x, 0.2, "euler", True, _noise())
    assert maxabs(st, sr) <= TOL and maxabs(inp, ir) <= TOL


def test_c32_config1_cell(pair8):
    """BASELINE config 1 state: 50x50x32."""
    net, sd = build_pair(32)
    x = hashfill.normal("x32", (1, 32, 50, 50), 31)
    s = hashfill.normal("s32", (1, 32, 50, 50), 32) * 0.5
    net.gru_ode.noise = _noise()
    st, inp, *_ = net.gru_ode.ode_step(s.cuda(), x.cuda(), 0.05, 0.0)
    with torch.no_grad():
        sr, ir = R.ode_step(sd, "gru_ode", s, x, 0.05, "euler", True, _noise())
    assert maxabs(st, sr) <= TOL and maxabs(inp, ir) <= TOL


def test_split_k_handoff_is_stable_under_repetition(pair64):
    """The 7x7 trusting-gate conv runs with its K range split over workgroups (slab publish +
    ticket + last-arriver reduce).  A stale read would show up as a run-to-run difference: 150
    back-to-back evaluations at 1 and 4 batched samples must be bitwise identical."""
    net, _ = pair64
    cell = net.gru_ode.gru_c
    one = torch.ones(1, device="cuda")
    for B in (1, 4):
        x = hashfill.normal("rx", (B, 50, 50, 64), 61).cuda()
        s = (hashfill.normal("rs", (B, 50, 50, 64), 62) * 0.5).cuda()
        ref = torch.empty_like(s)
        cell.run_nhwc(x, s, ref, True, s, one)
        ref = ref.clone()
        out = torch.empty_like(s)
        for i in range(150):
            cell.run_nhwc(x, s, out, True, s, one)
            if i % 10 == 9:
                assert torch.equal(out, ref), (B, i)


@pytest.mark.parametrize("B", [2, 3, 9])
def test_infer_state_of_batched_samples_equals_per_sample_calls(pair64, B):
    """Several 50x50 samples share one pixel space; the SE gates are per sample.  With 2500 pixels per image the 16-pixel
    groups whose channel sums feed the gates straddle the image borders (the producers split those sums at the border):
    the batched call must reproduce the per-sample calls (same kernels at B = 1 and the oracle-checked path)."""
    net, _ = pair64
    ode = net.gru_ode
    s = (hashfill.normal("bs_s", (B, 64, 50, 50), 71) * 0.5).cuda()
    eps = hashfill.normal("bs_e", (B, 64, 50, 50), 72)
    try:
        ode.noise = lambda shape, dtype, device: eps.clone()
        with torch.no_grad():
            y, q = ode.infer_state(s)
        for i in range(B):
            ode.noise = lambda shape, dtype, device, i=i: eps[i:i + 1].clone()
            with torch.no_grad():
                yi, qi = ode.infer_state(s[i:i + 1])
            assert maxabs(q[i:i + 1], qi) <= 2e-5 and maxabs(y[i:i + 1], yi) <= 2e-5, i
    finally:
        ode.noise = None


@pytest.mark.parametrize("C", [96, 128])
def test_hidden_sizes_above_64_vs_oracle(C):
    """Hidden sizes no shipped config uses (65..128 channels): the LayerNorm / trusting-gate epilogues then run on
    128-channel tiles (all channels of a pixel in one wave).  Dual cells and one Euler step against the oracle."""
    net, sd = build_pair(C)
    h, w = 16, 20
    x = hashfill.normal("xw", (1, C, h, w), 81)
    s = hashfill.normal("sw", (1, C, h, w), 82) * 0.5
    ode = net.gru_ode
    try:
        with torch.no_grad():
            assert maxabs(ode.gru_c(x.cuda(), s.cuda()), R.dual_cell(sd, "gru_ode.gru_c", x, s, True)) <= TOL
            assert maxabs(ode.gru_obs(s.cuda(), None, x.cuda())[0], R.dual_cell(sd, "gru_ode.gru_obs.gru_d", x, s, False)) <= TOL
            ode.noise = _noise()
            st, inp, *_ = ode.ode_step(s.cuda(), x.cuda(), 0.3, 0.0)
            sr, ir = R.ode_step(sd, "gru_ode", s, x, 0.3, "euler", True, _noise())
        assert maxabs(st, sr) <= TOL and maxabs(inp, ir) <= TOL
    finally:
        ode.noise = None


@pytest.mark.parametrize("C,h,w,B", [(8, 5, 7, 1), (16, 13, 9, 2), (24, 31, 17, 1), (32, 50, 50, 1), (48, 23, 40, 1), (64, 37, 41, 1),
                                     (64, 63, 64, 1), (64, 64, 65, 1), (40, 19, 21, 3), (64, 8, 8, 5)])
def test_cells_and_step_at_odd_latent_sizes(C, h, w, B):
    """Latent sizes and widths no config ships (ragged last tiles, 1..5 samples in one pixel space, channel counts that pad
    to the next 32): dual ODE cell, observation cell, infer_state and one Euler step against the oracle, sample by sample."""
    net, sd = build_pair(C)
    ode = net.gru_ode
    x = hashfill.normal("xo", (B, C, h, w), 91)
    s = hashfill.normal("so", (B, C, h, w), 92) * 0.5
    eps = hashfill.normal("eo", (B, C, h, w), 93)
    try:
        with torch.no_grad():
            d = ode.gru_c(x.cuda(), s.cuda())
            o = ode.gru_obs(s.cuda(), None, x.cuda())[0]
            ode.noise = lambda shape, dtype, device: eps.clone()
            st, inp, *_ = ode.ode_step(s.cuda(), x.cuda(), 0.25, 0.0)
            for i in range(B):
                xi, si = x[i:i + 1], s[i:i + 1]
                assert maxabs(d[i:i + 1], R.dual_cell(sd, "gru_ode.gru_c", xi, si, True)) <= TOL
                assert maxabs(o[i:i + 1], R.dual_cell(sd, "gru_ode.gru_obs.gru_d", xi, si, False)) <= TOL
                sr, ir = R.ode_step(sd, "gru_ode", si, xi, 0.25, "euler", True, lambda shape, dtype, device, i=i: eps[i:i + 1].clone())
                assert maxabs(st[i:i + 1], sr) <= TOL and maxabs(inp[i:i + 1], ir) <= TOL
    finally:
        ode.noise = None
