"""MI355X-native counterparts of streamingflow/layers/res_models.py: ``ConvBlock`` (:8-49),
``ResBlock`` (:52-79), ``SmallEncoder`` (:82-109), ``SmallDecoder`` (:112-147), ``SELayer``
(:150-165) and ``ConvNet`` (:168-180, the GRU-ODE p_model).

Same constructor signatures and ``state_dict`` keys as the reference; ``forward`` runs on
libsfnative (HIP, gfx950): eval-mode BatchNorm is folded into the conv epilogue, Dropout2d is inert,
ConvTranspose2d(k3,s1,p1) is executed as the equivalent convolution.  Inference only.
"""
from collections import OrderedDict

import torch
import torch.nn as nn

from .. import _lib, packing, runtime
from ..runtime import PackedModule, ptr

_ACTS = {"lrelu": nn.LeakyReLU, "relu": nn.ReLU, "tanh": nn.Tanh}


class ConvBlock(nn.Module):
    def __init__(self, in_channels, out_channels=None, kernel_size=3, stride=1, norm='bn', activation='lrelu',
                 bias=False, transpose=False):
        super().__init__()
        out_channels = out_channels or in_channels
        if kernel_size != 3 or stride != 1:
            raise NotImplementedError("hot path uses k3/s1 ConvBlocks only")
        ctor = nn.ConvTranspose2d if transpose else nn.Conv2d
        self.conv = ctor(in_channels, out_channels, kernel_size, stride, padding=1, bias=bias)
        if norm not in ("bn", "none"):
            raise NotImplementedError(norm)
        self.norm = nn.BatchNorm2d(out_channels) if norm == "bn" else None
        if activation not in ("lrelu", "relu", "tanh", "none"):
            raise ValueError('Invalid activation {}'.format(activation))
        self.activation = None if activation == "none" else (
            nn.LeakyReLU(0.1) if activation == "lrelu" else _ACTS[activation]())
        self.act_name, self.transpose, self.cin = activation, transpose, in_channels

    def pack(self, holder, interleave=False):
        if self.training and self.norm is not None:
            raise RuntimeError("streamingflow_amd is inference-only: call .eval() (BatchNorm uses running statistics)")
        scale, bias = (None, self.conv.bias) if self.norm is None else packing.bn_fold(self.norm, self.conv.bias)
        return packing.conv_w(holder, self.conv.weight, self.cin, scale=scale, bias=bias, act=self.act_name,
                              transposed=self.transpose, interleave=interleave)


class ResBlock(nn.Module):
    def __init__(self, in_channels, out_channels=None, norm='bn', activation='lrelu', bias=False):
        super().__init__()
        out_channels = out_channels or in_channels
        self.layers = nn.Sequential(OrderedDict([
            ('conv_1', ConvBlock(in_channels, in_channels, 3, 1, norm, activation, bias)),
            ('conv_2', ConvBlock(in_channels, out_channels, 3, 1, norm, activation, bias)),
            ('dropout', nn.Dropout2d(0.25))]))
        self.projection = nn.Conv2d(in_channels, out_channels, 1) if out_channels != in_channels else None
        self.cin = in_channels

    def pack(self, holder):
        s = _lib.ResW()
        s.conv1 = self.layers.conv_1.pack(holder)
        s.conv2 = self.layers.conv_2.pack(holder)
        if self.projection is not None:
            s.proj = packing.conv_w(holder, self.projection.weight, self.cin, bias=self.projection.bias)
        return s


class SmallEncoder(PackedModule):
    """[n, C, H, W] -> [n, nh, H/4, W/4] (two 2x2 max-pools, tanh output)."""

    def __init__(self, nc, nh, nf):
        super().__init__()
        self.blocks = nn.ModuleList([ResBlock(nc, nf), ResBlock(nf, nf * 2), ResBlock(nf * 2, nf * 2),
                                     ResBlock(nf * 2, nf * 2), ResBlock(nf * 2, nf * 4)])
        self.last_conv = nn.Sequential(ConvBlock(nf * 4, nh, 3, stride=1, activation='tanh'))
        self.maxpool = nn.MaxPool2d(kernel_size=2, stride=2, padding=0)
        self.nc, self.nh, self.nf = nc, nh, nf

    def _pack(self):
        pk = packing.Pack(_lib.EncoderW())
        for i, b in enumerate(self.blocks):
            pk.struct.blocks[i] = b.pack(pk)
        pk.struct.last = self.last_conv[0].pack(pk)
        pk.struct.C, pk.struct.F = self.nc, self.nf
        return pk

    def forward_nhwc(self, x):
        n, h, w, c = x.shape
        L = _lib.lib()
        ws = runtime.workspace(L.sf_small_encoder_ws_bytes(max(c, self.nh), self.nf, n, h, w), x.device)
        out = torch.empty((n, (h // 2) // 2, (w // 2) // 2, self.nh), dtype=torch.float32, device=x.device)
        _lib.check(L.sf_small_encoder_fwd(self.packed().struct, ptr(x), ptr(out), n, h, w, ptr(ws), ws.numel() * 4,
                                          runtime.stream_ptr(x.device)), "small_encoder")
        return out

    def forward(self, x, return_skip=False):
        if return_skip:
            raise NotImplementedError("skip connections (MODEL.SMALL_ENCODER.SKIPCO) are not on the shipped path")
        return runtime.to_nchw(self.forward_nhwc(runtime.to_nhwc(x)))


class SmallDecoder(PackedModule):
    """[n, nc, h, w] -> [n, nh, 4h, 4w] (two nearest x2 upsamplings)."""

    def __init__(self, nc, nh, nf, skip):
        super().__init__()
        if skip:
            raise NotImplementedError("skip connections (MODEL.SMALL_ENCODER.SKIPCO) are not on the shipped path")
        self.skip = skip
        self.first_upconv = ConvBlock(nc, nf * 4, stride=1, transpose=True)
        self.blocks = nn.ModuleList([ResBlock(nf * 4, nf * 2), ResBlock(nf * 2, nf * 2), ResBlock(nf * 2, nf * 2),
                                     ResBlock(nf * 2, nf), ResBlock(nf, nf)])
        self.last_conv = nn.Sequential(ConvBlock(nf, nf, 3, stride=1),
                                       ConvBlock(nf, nh, 3, stride=1, transpose=True, bias=True, norm='none'))
        self.upsample = nn.Upsample(scale_factor=2, mode='nearest')
        self.nc, self.nh, self.nf = nc, nh, nf

    def _pack(self):
        pk = packing.Pack(_lib.DecoderW())
        s = pk.struct
        s.first = self.first_upconv.pack(pk)
        for i, b in enumerate(self.blocks):
            s.blocks[i] = b.pack(pk)
        s.last0, s.last1 = self.last_conv[0].pack(pk), self.last_conv[1].pack(pk)
        s.C, s.F = self.nh, self.nf
        return pk

    def forward_nhwc(self, z):
        n, h, w, c = z.shape
        L = _lib.lib()
        ws = runtime.workspace(L.sf_small_decoder_ws_bytes(max(c, self.nh), self.nf, n, h, w), z.device)
        out = torch.empty((n, 4 * h, 4 * w, self.nh), dtype=torch.float32, device=z.device)
        _lib.check(L.sf_small_decoder_fwd(self.packed().struct, ptr(z), ptr(out), n, h, w, ptr(ws), ws.numel() * 4,
                                          runtime.stream_ptr(z.device)), "small_decoder")
        return out

    def forward(self, z, skip=None, sigmoid=False):
        assert skip is None and not self.skip or self.skip and skip is not None
        out = runtime.to_nchw(self.forward_nhwc(runtime.to_nhwc(z)))
        return torch.sigmoid(out) if sigmoid else out


class SELayer(nn.Module):
    """Parameter container (squeeze-excite gate, executed inside sf_infer_state_fwd)."""

    def __init__(self, channel, reduction=8):
        super().__init__()
        self.avg_pool = nn.AdaptiveAvgPool2d(1)
        self.fc = nn.Sequential(nn.Linear(channel, channel // reduction, bias=False), nn.ReLU(inplace=True),
                                nn.Linear(channel // reduction, channel, bias=False), nn.Sigmoid())


class ConvNet(PackedModule):
    """p_model: ResBlock -> SE -> ResBlock -> SE -> conv+LeakyReLU.  Executed (together with the
    Gaussian reparameterised sample) by NNFOwithBayesianJumps.infer_state."""

    def __init__(self, in_c, out_c):
        super().__init__()
        if out_c != 2 * in_c or in_c % 8:
            raise NotImplementedError("p_model is ConvNet(C, 2C) with C a multiple of 8")
        self.model = nn.Sequential(ResBlock(in_c, out_c), SELayer(out_c), ResBlock(out_c, out_c), SELayer(out_c),
                                   ConvBlock(out_c, out_c, 3, stride=1, bias=True, norm='none'))
        self.in_c = in_c

    def _pack(self):
        pk = packing.Pack(_lib.PModelW())
        s, m = pk.struct, self.model
        s.rb0, s.rb1 = m[0].pack(pk), m[2].pack(pk)
        s.se0_fc0, s.se0_fc2 = pk.hold(m[1].fc[0].weight), pk.hold(m[1].fc[2].weight)
        s.se1_fc0, s.se1_fc2 = pk.hold(m[3].fc[0].weight), pk.hold(m[3].fc[2].weight)
        s.last = m[4].pack(pk, interleave=True)
        s.C = self.in_c
        return pk
