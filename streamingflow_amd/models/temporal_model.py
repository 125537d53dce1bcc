"""``TemporalModel`` (streamingflow/models/temporal_model.py:8-69) on the MI355X conv library — SURVEY.md §8f N3.

Same constructor, ``forward`` and ``state_dict`` keys as the reference (``model.{i}.convolution_paths...``,
``pyramid_pooling.features.0.conv_bn_relu``, ``aggregation.0``, ``projection``, ``final_conv`` = DeepLabHead).
The 3-D convolutions of a ``TemporalBlock`` (layers/temporal.py:435-490) are causal in time with
kernels (2,3,3) / (1,3,3) / (1,1,1), so every one of them is a 2-D convolution over frames:

* the three leading 1x1x1 convs share their input: two are one stacked 1x1 launch, the third writes
  straight into its slice of the concat tensor;
* the (2,3,3) causal conv of frame t is a 3x3 conv over the channel concat [frame t-1 | frame t]
  (a zero frame in front of the sequence plays the temporal padding);
* PyramidSpatioTemporalPooling with pool size (2, h, w): per-frame channel means (library kernel),
  averaged over (t-1, t) as AvgPool3d(count_include_pad=False) does, 1x1x1 conv on the vectors, and
  the bilinear upsampling of a 1x1 map is a broadcast into the concat tensor;
* aggregation 1x1x1 + BN + ReLU with the (projected) input added in the epilogue.

Channel counts that are not multiples of 4 (70 -> 35 / 23 in the camera branch) are zero-padded in the
packed weights.  Evaluation mode only, CUDA tensors only (no CPU fallback).
"""
from collections import OrderedDict

import torch
import torch.nn as nn

from .. import _lib, packing, runtime
from ..layers.convolutions import DeepLabHead
from ..runtime import PackedModule, ptr
from .decoder import C_void


def conv_1x1x1_norm_activated(in_channels, out_channels):
    return nn.Sequential(OrderedDict([("conv", nn.Conv3d(in_channels, out_channels, kernel_size=1, bias=False)),
                                      ("norm", nn.BatchNorm3d(out_channels)), ("activation", nn.ReLU(inplace=True))]))


class CausalConv3d(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=(2, 3, 3)):
        super().__init__()
        self.kernel_size = kernel_size
        self.conv = nn.Conv3d(in_channels, out_channels, kernel_size, stride=1, padding=0, bias=False)
        self.norm = nn.BatchNorm3d(out_channels)
        self.activation = nn.ReLU(inplace=True)


class PyramidSpatioTemporalPooling(nn.Module):
    def __init__(self, in_channels, reduction_channels, pool_sizes):
        super().__init__()
        assert len(pool_sizes) == 1 and pool_sizes[0][0] == 2
        self.pool_sizes = pool_sizes
        self.features = nn.ModuleList([nn.Sequential(OrderedDict([
            ("avgpool", nn.AvgPool3d(kernel_size=ps, stride=(1, *ps[1:]), padding=(ps[0] - 1, 0, 0), count_include_pad=False)),
            ("conv_bn_relu", conv_1x1x1_norm_activated(in_channels, reduction_channels))])) for ps in pool_sizes])


class TemporalBlock(nn.Module):
    def __init__(self, in_channels, out_channels=None, use_pyramid_pooling=False, pool_sizes=None):
        super().__init__()
        self.in_channels, self.half_channels = in_channels, in_channels // 2
        self.out_channels = out_channels or in_channels
        self.use_pyramid_pooling = use_pyramid_pooling
        paths = [nn.Sequential(conv_1x1x1_norm_activated(in_channels, self.half_channels),
                               CausalConv3d(self.half_channels, self.half_channels, kernel_size=k)) for k in ((2, 3, 3), (1, 3, 3))]
        paths.append(conv_1x1x1_norm_activated(in_channels, self.half_channels))
        self.convolution_paths = nn.ModuleList(paths)
        agg_in = 3 * self.half_channels
        if use_pyramid_pooling:
            assert pool_sizes is not None
            self.reduction_channels = in_channels // 3
            self.pyramid_pooling = PyramidSpatioTemporalPooling(in_channels, self.reduction_channels, pool_sizes)
            agg_in += len(pool_sizes) * self.reduction_channels
        self.aggregation = nn.Sequential(conv_1x1x1_norm_activated(agg_in, self.out_channels))
        self.projection = None
        if self.out_channels != in_channels:
            self.projection = nn.Sequential(nn.Conv3d(in_channels, self.out_channels, kernel_size=1, bias=False),
                                            nn.BatchNorm3d(self.out_channels))


class Bottleneck3D(nn.Module):
    """layers/temporal.py:333-391: 1x1x1 down -> causal conv -> 1x1x1 up, residual (projected when channels change)."""

    def __init__(self, in_channels, out_channels=None, kernel_size=(2, 3, 3), dilation=(1, 1, 1)):
        super().__init__()
        if tuple(dilation) != (1, 1, 1) or tuple(kernel_size) not in ((2, 3, 3), (1, 3, 3)):
            raise NotImplementedError("Bottleneck3D: kernels (2,3,3) / (1,3,3), dilation 1")
        bottleneck_channels = in_channels // 2
        out_channels = out_channels or in_channels
        self.in_channels, self.out_channels, self.bottleneck_channels = in_channels, out_channels, bottleneck_channels
        self.layers = nn.Sequential(OrderedDict([
            ("conv_down_project", conv_1x1x1_norm_activated(in_channels, bottleneck_channels)),
            ("conv", CausalConv3d(bottleneck_channels, bottleneck_channels, kernel_size=kernel_size)),
            ("conv_up_project", conv_1x1x1_norm_activated(bottleneck_channels, out_channels))]))
        self.projection = None
        if out_channels != in_channels:
            self.projection = nn.Sequential(nn.Conv3d(in_channels, out_channels, kernel_size=1, bias=False), nn.BatchNorm3d(out_channels))


def _p4(c):
    return (c + 3) // 4 * 4


class TemporalModel(PackedModule):
    def __init__(self, in_channels, receptive_field, input_shape, start_out_channels=64, extra_in_channels=0,
                 n_spatial_layers_between_temporal_layers=0, use_pyramid_pooling=True, with_cp=False):
        super().__init__()
        self.receptive_field = receptive_field
        h, w = input_shape
        self.input_shape = (int(h), int(w))
        modules = []
        cin, cout = in_channels, start_out_channels
        for _ in range(receptive_field - 1):
            modules.append(TemporalBlock(cin, cout, use_pyramid_pooling=bool(use_pyramid_pooling),
                                         pool_sizes=[(2, h, w)] if use_pyramid_pooling else None))
            modules.extend(Bottleneck3D(cout, cout, kernel_size=(1, 3, 3)) for _ in range(n_spatial_layers_between_temporal_layers))
            cin = cout
            cout += extra_in_channels
        self.out_channels = cin
        self.final_conv = DeepLabHead(cout, cout, hidden_channel=128)
        self.model = nn.Sequential(*modules)
        self.with_cp = with_cp
        self.in_channels = in_channels

    # ---- packing ----------------------------------------------------------------------------------
    def _pack(self):
        pk = packing.Pack([])
        for blk in self.model:
            if isinstance(blk, Bottleneck3D):
                pk.struct.append(self._pack_bottleneck(pk, blk))
                continue
            C, Ch, Co = blk.in_channels, blk.half_channels, blk.out_channels
            Cp, Chp = _p4(C), _p4(Ch)
            dev = blk.aggregation[0].conv.weight.device
            W = {"C": C, "Cp": Cp, "Ch": Ch, "Chp": Chp, "Co": Co}

            def lead(seq):          # 1x1x1 conv + BN + ReLU on the (zero-padded) input channels
                sc, bi = packing.bn_fold(seq.norm)
                w = torch.zeros((Chp, Cp, 1, 1), device=dev)
                w[:Ch, :C] = seq.conv.weight.detach()[:, :, 0, 0, 0][:, :, None, None]
                s, b = torch.zeros(Chp, device=dev), torch.zeros(Chp, device=dev)
                s[:Ch], b[:Ch] = sc, bi
                return w, s, b
            w0, s0, b0 = lead(blk.convolution_paths[0][0])
            w1, s1, b1 = lead(blk.convolution_paths[1][0])
            w2, s2, b2 = lead(blk.convolution_paths[2])
            W["a01"] = packing.conv_w(pk, torch.cat([w0, w1], 0), Cp, 0, torch.cat([s0, s1]), torch.cat([b0, b1]), "relu", pad=0)
            W["a2"] = packing.conv_w(pk, w2, Cp, 0, s2, b2, "relu", pad=0)
            # causal (2,3,3): [prev frame | current frame] channel concat, 3x3
            c0 = blk.convolution_paths[0][1]
            sc, bi = packing.bn_fold(c0.norm)
            w3 = c0.conv.weight.detach()                                   # [Ch][Ch][2][3][3]
            w = torch.zeros((Chp, 2 * Chp, 3, 3), device=dev)
            w[:Ch, :Ch], w[:Ch, Chp:Chp + Ch] = w3[:, :, 0], w3[:, :, 1]
            s, b = torch.zeros(Chp, device=dev), torch.zeros(Chp, device=dev)
            s[:Ch], b[:Ch] = sc, bi
            W["p0"] = packing.conv_w(pk, w, Chp, Chp, s, b, "relu", pad=1)
            c1 = blk.convolution_paths[1][1]
            sc, bi = packing.bn_fold(c1.norm)
            w = torch.zeros((Chp, Chp, 3, 3), device=dev)
            w[:Ch, :Ch] = c1.conv.weight.detach()[:, :, 0]
            s, b = torch.zeros(Chp, device=dev), torch.zeros(Chp, device=dev)
            s[:Ch], b[:Ch] = sc, bi
            W["p1"] = packing.conv_w(pk, w, Chp, 0, s, b, "relu", pad=1)
            cols = [(0, Ch, 0), (Ch, Ch, Chp), (2 * Ch, Ch, 2 * Chp)]       # (source column, count, packed column)
            Wc = 3 * Chp
            if blk.use_pyramid_pooling:
                Cr = blk.reduction_channels
                Crp = _p4(Cr)
                f = blk.pyramid_pooling.features[0].conv_bn_relu
                sc, bi = packing.bn_fold(f.norm)
                w = torch.zeros((Crp, Cp, 1, 1), device=dev)
                w[:Cr, :C] = f.conv.weight.detach()[:, :, 0, 0, 0][:, :, None, None]
                s, b = torch.zeros(Crp, device=dev), torch.zeros(Crp, device=dev)
                s[:Cr], b[:Cr] = sc, bi
                W["pool"] = packing.conv_w(pk, w, Cp, 0, s, b, "relu", pad=0)
                W["Crp"] = Crp
                cols.append((3 * Ch, Cr, 3 * Chp))
                Wc += Crp
            W["Wc"] = Wc
            agg = blk.aggregation[0]
            sc, bi = packing.bn_fold(agg.norm)
            wa = agg.conv.weight.detach()[:, :, 0, 0, 0]
            w = torch.zeros((Co, Wc, 1, 1), device=dev)
            for src, cnt, dst in cols:
                w[:, dst:dst + cnt, 0, 0] = wa[:, src:src + cnt]
            W["agg"] = packing.conv_w(pk, w, Wc, 0, sc, bi, "relu", pad=0)
            if blk.projection is not None:
                sc, bi = packing.bn_fold(blk.projection[1])
                w = torch.zeros((Co, Cp, 1, 1), device=dev)
                w[:, :C] = blk.projection[0].weight.detach()[:, :, 0, 0, 0][:, :, None, None]
                W["proj"] = packing.conv_w(pk, w, Cp, 0, sc, bi, "none", pad=0)
            elif Cp != Co:
                raise RuntimeError("TemporalBlock without projection needs in_channels == out_channels (multiple of 4)")
            pk.struct.append(W)
        return pk

    @staticmethod
    def _pack_bottleneck(pk, blk):
        C, Cb, Co = blk.in_channels, blk.bottleneck_channels, blk.out_channels
        Cp, Cbp = _p4(C), _p4(Cb)
        if Co % 4:
            raise RuntimeError("Bottleneck3D out_channels must be a multiple of 4")
        dev = blk.layers.conv.conv.weight.device
        W = {"kind": "bottleneck", "Cp": Cp, "Cbp": Cbp, "Co": Co}

        def fold(bn, n, npad):
            sc, bi = packing.bn_fold(bn)
            s, b = torch.zeros(npad, device=dev), torch.zeros(npad, device=dev)
            s[:n], b[:n] = sc, bi
            return s, b
        d = blk.layers.conv_down_project
        w = torch.zeros((Cbp, Cp, 1, 1), device=dev)
        w[:Cb, :C] = d.conv.weight.detach()[:, :, 0, 0, 0][:, :, None, None]
        W["down"] = packing.conv_w(pk, w, Cp, 0, *fold(d.norm, Cb, Cbp), "relu", pad=0)
        c = blk.layers.conv
        w3 = c.conv.weight.detach()                                       # [Cb][Cb][kt][3][3]
        kt = w3.shape[2]
        W["kt"] = kt
        w = torch.zeros((Cbp, kt * Cbp, 3, 3), device=dev)
        for t in range(kt):
            w[:Cb, t * Cbp:t * Cbp + Cb] = w3[:, :, t]
        W["conv"] = packing.conv_w(pk, w, Cbp, Cbp if kt == 2 else 0, *fold(c.norm, Cb, Cbp), "relu", pad=1)
        u = blk.layers.conv_up_project
        w = torch.zeros((Co, Cbp, 1, 1), device=dev)
        w[:, :Cb] = u.conv.weight.detach()[:, :, 0, 0, 0][:, :, None, None]
        sc, bi = packing.bn_fold(u.norm)
        W["up"] = packing.conv_w(pk, w, Cbp, 0, sc, bi, "relu", pad=0)
        if blk.projection is not None:
            sc, bi = packing.bn_fold(blk.projection[1])
            w = torch.zeros((Co, Cp, 1, 1), device=dev)
            w[:, :C] = blk.projection[0].weight.detach()[:, :, 0, 0, 0][:, :, None, None]
            W["proj"] = packing.conv_w(pk, w, Cp, 0, sc, bi, "none", pad=0)
        elif Cp != Co:
            raise RuntimeError("Bottleneck3D without projection needs in_channels == out_channels (multiple of 4)")
        return W

    def _bottleneck(self, W, x, b, T, H, Wd):
        """x [b*T, H, W, Cp] -> [b*T, H, W, Co]   (layers/temporal.py:386-391)."""
        dev = x.device
        n = b * T
        Cp, Cbp, Co = W["Cp"], W["Cbp"], W["Co"]
        A = torch.zeros((b, T + 1, H, Wd, Cbp), dtype=torch.float32, device=dev)      # one zero frame = the causal time padding
        for bi in range(b):
            self._conv(W["down"], x[bi * T:], Cp, 0, T, H, Wd, A[bi, 1:], Cbp, 0)
        y = torch.empty((n, H, Wd, Cbp), dtype=torch.float32, device=dev)
        for bi in range(b):
            if W["kt"] == 2:
                self._conv(W["conv"], A[bi, :T], Cbp, 0, T, H, Wd, y[bi * T:], Cbp, 0, in1=A[bi, 1:], in1_cs=Cbp, in1_co=0)
            else:
                self._conv(W["conv"], A[bi, 1:], Cbp, 0, T, H, Wd, y[bi * T:], Cbp, 0)
        if "proj" in W:
            res = torch.empty((n, H, Wd, Co), dtype=torch.float32, device=dev)
            self._conv(W["proj"], x, Cp, 0, n, H, Wd, res, Co, 0)
        else:
            res = x
        out = torch.empty((n, H, Wd, Co), dtype=torch.float32, device=dev)
        self._conv(W["up"], y, Cbp, 0, n, H, Wd, out, Co, 0, add=res, add_cs=Co)
        return out

    # ---- device side ------------------------------------------------------------------------------
    @staticmethod
    def _conv(w, in0, in0_cs, in0_co, n, H, Wd, out, out_cs, out_co, in1=None, in1_cs=0, in1_co=0, add=None, add_cs=0):
        L = _lib.lib()
        dev = out.device
        ws = runtime.workspace(L.sf_conv2d_ex_ws_bytes(), dev)
        _lib.check(L.sf_conv2d_ex_fwd(_lib.C.byref(w), C_void(in0, in0_co), in0_cs, C_void(in1, in1_co) if in1 is not None else None,
                                      in1_cs, ptr(add), add_cs, 0, ptr(out), out_cs, out_co, n, H, Wd, 0, ptr(ws), ws.numel() * 4,
                                      runtime.stream_ptr(dev)), "conv2d_ex")

    def _block(self, W, x, b, T, H, Wd):
        """x [b*T, H, W, Cp] -> [b*T, H, W, Co]."""
        dev = x.device
        L = _lib.lib()
        st = runtime.stream_ptr(dev)
        n = b * T
        Cp, Chp, Co, Wc = W["Cp"], W["Chp"], W["Co"], W["Wc"]
        # leading 1x1x1 convs: (a0 | a1) with one zero frame in front of every sample, a2 into the concat tensor
        A = torch.zeros((b, T + 1, H, Wd, 2 * Chp), dtype=torch.float32, device=dev)
        cat = torch.empty((n, H, Wd, Wc), dtype=torch.float32, device=dev)
        for bi in range(b):
            self._conv(W["a01"], x[bi * T:], Cp, 0, T, H, Wd, A[bi, 1:], 2 * Chp, 0)
        self._conv(W["a2"], x, Cp, 0, n, H, Wd, cat, Wc, 2 * Chp)
        for bi in range(b):       # causal (2,3,3): frames t-1 (A[bi, t]) and t (A[bi, t+1]), channel slice a0
            self._conv(W["p0"], A[bi, :T], 2 * Chp, 0, T, H, Wd, cat[bi * T:], Wc, 0, in1=A[bi, 1:], in1_cs=2 * Chp, in1_co=0)
            self._conv(W["p1"], A[bi, 1:], 2 * Chp, Chp, T, H, Wd, cat[bi * T:], Wc, Chp)
        if "pool" in W:
            Crp = W["Crp"]
            mean = torch.empty((n, Cp), dtype=torch.float32, device=dev)
            wsm = runtime.workspace(L.sf_channel_mean_ws_bytes(Cp, n), dev)
            _lib.check(L.sf_channel_mean_fwd(ptr(x), ptr(mean), n, H * Wd, Cp, ptr(wsm), wsm.numel() * 4, st), "channel_mean")
            m = mean.view(b, T, Cp)
            pooled = m.clone()
            pooled[:, 1:] = (m[:, :-1] + m[:, 1:]) * 0.5         # AvgPool3d((2,h,w), pad (1,0,0), count_include_pad=False)[:-1]
            vec = torch.empty((n, Crp), dtype=torch.float32, device=dev)
            self._conv(W["pool"], pooled.view(n, Cp), Cp, 0, n, 1, 1, vec, Crp, 0)
            _lib.check(L.sf_broadcast_channels_fwd(ptr(vec), ptr(cat), n, H * Wd, Crp, Wc, 3 * Chp, st), "broadcast")
        if "proj" in W:
            res = torch.empty((n, H, Wd, Co), dtype=torch.float32, device=dev)
            self._conv(W["proj"], x, Cp, 0, n, H, Wd, res, Co, 0)
        else:
            res = x
        out = torch.empty((n, H, Wd, Co), dtype=torch.float32, device=dev)
        self._conv(W["agg"], cat, Wc, 0, n, H, Wd, out, Co, 0, add=res, add_cs=Co)
        return out

    def forward(self, x):
        """x [b, s, c, h, w] -> [b, s, c_out, h, w] (temporal_model.py:51-69)."""
        runtime.require_cuda(x)
        if self.training:
            raise RuntimeError("streamingflow_amd.TemporalModel is inference-only (BatchNorm statistics are folded): call .eval()")
        b, s, c, h, w = x.shape
        if len(self.model) and getattr(self.model[0], "use_pyramid_pooling", False) and (h, w) != self.input_shape:
            raise RuntimeError(f"pyramid pooling was built for a {self.input_shape} grid, got {(h, w)}")
        packs = self.packed().struct
        Cp = _p4(c)
        xf = x.reshape(b * s, c, h, w).float()
        if Cp != c:
            xf = torch.nn.functional.pad(xf, (0, 0, 0, 0, 0, Cp - c))
        y = runtime.to_nhwc(xf)
        for W in packs:
            y = self._bottleneck(W, y, b, s, h, w) if W.get("kind") == "bottleneck" else self._block(W, y, b, s, h, w)
        y = self.final_conv.forward_nhwc(y)
        return runtime.to_nchw(y).view(b, s, -1, h, w)
