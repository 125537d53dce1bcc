"""BEV ``Decoder`` (streamingflow/models/decoder.py:8-140) on the MI355X conv library — SURVEY.md §8f N3.

Same constructor, ``forward`` signature, output dict and ``state_dict`` keys as the reference
(including the torchvision ResNet-18 names ``layer{1,2,3}.{0,1}.{conv1,bn1,conv2,bn2,downsample.{0,1}}``).
Every convolution runs through ``sf_conv2d_ex_fwd`` (implicit-GEMM fp32 MFMA, BatchNorm folded into
the epilogue, residual add + ReLU fused); differences in *how* the same function is computed:

* ``UpsamplingAdd`` (convolutions.py:204-215): the 1x1 conv + BatchNorm run BEFORE the bilinear x2
  interpolation (they commute with it: the interpolation weights sum to one), at a quarter of the
  pixels; interpolation and skip add are one kernel (``sf_upsample_bilinear2_add_fwd``).
* the first 3x3 conv of all heads is ONE convolution with stacked output channels; each head's 1x1
  reads its channel slice of that tensor.

Evaluation mode only (BatchNorm running statistics are folded); CPU tensors raise — no fallback.
"""
import torch
import torch.nn as nn

from .. import _lib, packing, runtime
from ..runtime import PackedModule, ptr


class BasicBlock(nn.Module):
    """Parameter container with torchvision's BasicBlock names (computed by Decoder._run)."""

    def __init__(self, inplanes, planes, stride=1):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=1, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = None
        if stride != 1 or inplanes != planes:
            self.downsample = nn.Sequential(nn.Conv2d(inplanes, planes, 1, stride=stride, bias=False), nn.BatchNorm2d(planes))
        self.stride = stride


class UpsamplingAdd(nn.Module):
    def __init__(self, in_channels, out_channels, scale_factor=2):
        super().__init__()
        assert scale_factor == 2
        self.upsample_layer = nn.Sequential(nn.Upsample(scale_factor=scale_factor, mode="bilinear", align_corners=False),
                                            nn.Conv2d(in_channels, out_channels, kernel_size=1, padding=0, bias=False),
                                            nn.BatchNorm2d(out_channels))


def _head(cin, cout, sigmoid=False):
    layers = [nn.Conv2d(cin, cin, kernel_size=3, padding=1, bias=False), nn.BatchNorm2d(cin), nn.ReLU(inplace=True),
              nn.Conv2d(cin, cout, kernel_size=1, padding=0)]
    if sigmoid:
        layers.append(nn.Sigmoid())
    return nn.Sequential(*layers)


# output key, attribute, gate key (None: always), 1x1 output channels (callable of the ctor args), sigmoid
_HEADS = (("segmentation", "segmentation_head", None, lambda nc, nh: nc, False),
          ("pedestrian", "pedestrian_head", "predict_pedestrian", lambda nc, nh: nc, False),
          ("hdmap", "hdmap_head", "perceive_hdmap", lambda nc, nh: 2 * nh, False),
          ("instance_offset", "instance_offset_head", "predict_instance", lambda nc, nh: 2, False),
          ("instance_center", "instance_center_head", "predict_instance", lambda nc, nh: 1, True),
          ("instance_flow", "instance_future_head", "predict_future_flow", lambda nc, nh: 2, False),
          ("costvolume", "costvolume_head", "planning", lambda nc, nh: 1, False))


class Decoder(PackedModule):
    def __init__(self, in_channels, n_classes, n_present, n_hdmap, predict_gate):
        super().__init__()
        self.perceive_hdmap = predict_gate["perceive_hdmap"]
        self.predict_pedestrian = predict_gate["predict_pedestrian"]
        self.predict_instance = predict_gate["predict_instance"]
        self.predict_future_flow = predict_gate["predict_future_flow"]
        self.planning = predict_gate["planning"]
        self.n_classes, self.n_present, self.in_channels = n_classes, n_present, in_channels
        if self.predict_instance is False and self.predict_future_flow is True:
            raise ValueError("flow cannot be True when not predicting instance")
        if in_channels % 4:
            raise ValueError("in_channels must be a multiple of 4")
        self.first_conv = nn.Conv2d(in_channels, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.layer1 = nn.Sequential(BasicBlock(64, 64), BasicBlock(64, 64))
        self.layer2 = nn.Sequential(BasicBlock(64, 128, 2), BasicBlock(128, 128))
        self.layer3 = nn.Sequential(BasicBlock(128, 256, 2), BasicBlock(256, 256))
        for m in self.modules():                      # resnet18(zero_init_residual=True), decoder.py:22
            if isinstance(m, BasicBlock):
                nn.init.constant_(m.bn2.weight, 0)
        self.up3_skip = UpsamplingAdd(256, 128, scale_factor=2)
        self.up2_skip = UpsamplingAdd(128, 64, scale_factor=2)
        self.up1_skip = UpsamplingAdd(64, in_channels, scale_factor=2)
        gates = dict(predict_gate)
        self._heads = []
        for key, attr, gate, cout, sig in _HEADS:     # same construction order as decoder.py:38-91
            if gate is None or gates[gate]:
                setattr(self, attr, _head(in_channels, cout(n_classes, n_hdmap), sig))
                self._heads.append((key, attr, sig))

    # ---- packing ----------------------------------------------------------------------------------
    def _pack(self):
        pk = packing.Pack({})
        W = pk.struct

        def cbn(name, conv, bn, act, stride=1, pad=None):
            sc, bi = packing.bn_fold(bn)
            W[name] = packing.conv_w(pk, conv.weight, conv.in_channels, 0, sc, bi, act, stride=stride, pad=pad)
        cbn("first", self.first_conv, self.bn1, "relu", stride=2, pad=3)
        for ln, layer in (("layer1", self.layer1), ("layer2", self.layer2), ("layer3", self.layer3)):
            for i, blk in enumerate(layer):
                cbn(f"{ln}.{i}.c1", blk.conv1, blk.bn1, "relu", stride=blk.stride, pad=1)
                cbn(f"{ln}.{i}.c2", blk.conv2, blk.bn2, "relu", pad=1)             # relu applied after the residual add
                if blk.downsample is not None:
                    cbn(f"{ln}.{i}.down", blk.downsample[0], blk.downsample[1], "none", stride=blk.stride, pad=0)
        for name, up in (("up3", self.up3_skip), ("up2", self.up2_skip), ("up1", self.up1_skip)):
            cbn(name, up.upsample_layer[1], up.upsample_layer[2], "none", pad=0)
        # heads: stacked 3x3 (+BN+ReLU), then one 1x1 (+bias) per head on its channel slice
        C = self.in_channels
        ws, scs, bis = [], [], []
        for key, attr, sig in self._heads:
            h = getattr(self, attr)
            sc, bi = packing.bn_fold(h[1])
            ws.append(h[0].weight.detach()); scs.append(sc); bis.append(bi)
        W["heads3"] = packing.conv_w(pk, torch.cat(ws, 0), C, 0, torch.cat(scs), torch.cat(bis), "relu", pad=1)
        for key, attr, sig in self._heads:
            conv = getattr(self, attr)[3]
            k = conv.out_channels
            k4 = (k + 3) // 4 * 4                           # the library stores 4 channels at a time: zero-padded rows
            w = torch.zeros((k4, C, 1, 1), dtype=torch.float32, device=conv.weight.device)
            b = torch.zeros((k4,), dtype=torch.float32, device=conv.weight.device)
            w[:k], b[:k] = conv.weight.detach(), conv.bias.detach()
            W["head1." + key] = packing.conv_w(pk, w, C, 0, None, b, "sigmoid" if sig else "none", pad=0)
        return pk

    # ---- device helpers -----------------------------------------------------------------------------
    @staticmethod
    def _conv(w, x, n, H, W_, in_cs=None, in_co=0, add=None, act_after_add=False, out=None):
        L = _lib.lib()
        dev = x.device
        Ho = (H + 2 * w.pad - w.dil * (w.kh - 1) - 1) // w.stride + 1
        Wo = (W_ + 2 * w.pad - w.dil * (w.kw - 1) - 1) // w.stride + 1
        if out is None:
            out = torch.empty((n, Ho, Wo, w.cout), dtype=torch.float32, device=dev)
        ws = runtime.workspace(L.sf_conv2d_ex_ws_bytes(), dev)
        cs = in_cs if in_cs is not None else w.c0
        xin = C_void(x, in_co)
        _lib.check(L.sf_conv2d_ex_fwd(_lib.C.byref(w), xin, cs, None, 0, ptr(add), w.cout if add is not None else 0,
                                      int(act_after_add), ptr(out), w.cout, 0, n, H, W_, 0, ptr(ws), ws.numel() * 4,
                                      runtime.stream_ptr(dev)), "conv2d_ex")
        return out, Ho, Wo

    def _run(self, x):
        """x [n, H, W, C] NHWC -> (shared feature map [n, H, W, C], dict key -> [n, H, W, k4] head outputs)."""
        W = self.packed().struct
        n, H, Wd, C = x.shape
        L = _lib.lib()
        st = runtime.stream_ptr(x.device)
        y, h, w = self._conv(W["first"], x, n, H, Wd)
        skips = {1: (x, H, Wd)}
        for ln, layer, tag in (("layer1", self.layer1, 2), ("layer2", self.layer2, 3), ("layer3", self.layer3, None)):
            for i, blk in enumerate(layer):
                t, h2, w2 = self._conv(W[f"{ln}.{i}.c1"], y, n, h, w)
                idt = y
                if blk.downsample is not None:
                    idt, _, _ = self._conv(W[f"{ln}.{i}.down"], y, n, h, w)
                y, h, w = self._conv(W[f"{ln}.{i}.c2"], t, n, h2, w2, add=idt, act_after_add=True)
            if tag is not None:
                skips[tag] = (y, h, w)
        for name, tag in (("up3", 3), ("up2", 2), ("up1", 1)):
            sk, hs, ws_ = skips[tag]
            if (hs, ws_) != (2 * h, 2 * w):
                raise RuntimeError("Decoder needs a BEV size divisible by 8 (UpsamplingAdd x2 must meet its skip)")
            lo, _, _ = self._conv(W[name], y, n, h, w)
            y = torch.empty_like(sk)
            _lib.check(L.sf_upsample_bilinear2_add_fwd(ptr(lo), ptr(sk), ptr(y), n, h, w, lo.shape[-1], st), "upsample_add")
            h, w = hs, ws_
        return y

    def forward(self, x):
        """x [b, s, c, h, w] -> dict as decoder.py:120-140 (absent heads: None)."""
        runtime.require_cuda(x)
        if self.training:
            raise RuntimeError("streamingflow_amd.Decoder is inference-only (BatchNorm statistics are folded): call .eval()")
        b, s, c, h, w = x.shape
        xn = runtime.to_nhwc(x.reshape(b * s, c, h, w))
        W = self.packed().struct
        feat = self._run(xn)
        n = b * s
        C = self.in_channels
        stacked, _, _ = self._conv(W["heads3"], feat, n, h, w)                    # [n, h, w, C * n_heads]
        nh = len(self._heads)
        out = {k: None for k, _, _, _, _ in _HEADS}
        for i, (key, attr, sig) in enumerate(self._heads):
            w1 = W["head1." + key]
            k = getattr(self, attr)[3].out_channels
            if key == "hdmap":       # decoder.py:122: only the present frame
                sel = stacked.view(b, s, h, w, C * nh)[:, self.n_present - 1].contiguous()
                y, _, _ = self._conv(w1, sel, b, h, w, in_cs=C * nh, in_co=i * C)
                out[key] = runtime.to_nchw(y)[:, :k].contiguous()
                continue
            y, _, _ = self._conv(w1, stacked, n, h, w, in_cs=C * nh, in_co=i * C)
            y = runtime.to_nchw(y)[:, :k].contiguous()
            if key == "costvolume":
                y = y.squeeze(1)
            out[key] = y.view(b, s, *y.shape[1:])
        return out


def C_void(t, chan_offset=0):
    """Pointer to channel `chan_offset` of an NHWC tensor."""
    return _lib.C.c_void_p(t.data_ptr() + 4 * int(chan_offset))
