"""MI355X-native counterparts of the blocks of streamingflow/layers/convolutions.py that sit on the
GRU-ODE hot path: channels-first ``LayerNorm`` (:283-308), ConvNeXt ``Block`` (:310-346),
``Bottleblock`` (:348-380, the trusting-gate body) and ``ASPP``/``DeepLabHead`` (:217-280).

The classes keep the reference constructor signatures and ``state_dict`` keys (torch.nn layers are
used as parameter containers only); ``forward`` runs on libsfnative (HIP, gfx950).  Inference only.
"""
import torch
import torch.nn as nn

from .. import _lib, packing, runtime
from ..runtime import PackedModule, ptr


def _seq(*mods):
    return nn.Sequential(*mods)


class Bottleneck(PackedModule):
    """1x1 -> 3x3 (optionally stride 2) -> 1x1 bottleneck with a residual connection
    (convolutions.py:65-172; identical to BEVerse's basic_modules.Bottleneck)."""

    def __init__(self, in_channels, out_channels=None, kernel_size=3, dilation=1, groups=1, upsample=False,
                 downsample=False, dropout=0.0):
        super().__init__()
        if upsample or dilation != 1 or groups != 1 or kernel_size != 3:
            raise NotImplementedError("Bottleneck: only the k3 / dilation 1 / groups 1, non-upsampling form is on the path")
        from collections import OrderedDict
        self._downsample = downsample
        mid = int(in_channels / 2)
        out_channels = out_channels or in_channels

        def abn(c):
            return nn.Sequential(nn.BatchNorm2d(c), nn.ReLU(inplace=True))
        self.layers = nn.Sequential(OrderedDict([
            ('conv_down_project', nn.Conv2d(in_channels, mid, kernel_size=1, bias=False)), ('abn_down_project', abn(mid)),
            ('conv', nn.Conv2d(mid, mid, kernel_size=3, bias=False, stride=2 if downsample else 1, padding=1)),
            ('abn', abn(mid)),
            ('conv_up_project', nn.Conv2d(mid, out_channels, kernel_size=1, bias=False)), ('abn_up_project', abn(out_channels)),
            ('dropout', nn.Dropout2d(p=dropout))]))
        if out_channels == in_channels and not downsample:
            self.projection = None
        else:
            proj = OrderedDict()
            if downsample:
                proj['upsample_skip_proj'] = nn.MaxPool2d(kernel_size=2, stride=2)
            proj['conv_skip_proj'] = nn.Conv2d(in_channels, out_channels, kernel_size=1, bias=False)
            proj['bn_skip_proj'] = nn.BatchNorm2d(out_channels)
            self.projection = nn.Sequential(proj)
        self.cin, self.cout = in_channels, out_channels

    def _pack(self):
        if self.training:
            raise RuntimeError("streamingflow_amd is inference-only: call .eval() (BatchNorm uses running statistics)")
        pk = packing.Pack(_lib.BottleneckW())
        s, L = pk.struct, self.layers
        sc, bi = packing.bn_fold(L.abn_down_project[0])
        s.down = packing.conv_w(pk, L.conv_down_project.weight, self.cin, scale=sc, bias=bi, act="relu")
        sc, bi = packing.bn_fold(L.abn[0])
        s.conv = packing.conv_w(pk, L.conv.weight, self.cin // 2, scale=sc, bias=bi, act="relu",
                                stride=2 if self._downsample else 1, pad=1)
        sc, bi = packing.bn_fold(L.abn_up_project[0])
        s.up = packing.conv_w(pk, L.conv_up_project.weight, self.cin // 2, scale=sc, bias=bi, act="relu")
        if self.projection is not None:
            sc, bi = packing.bn_fold(self.projection.bn_skip_proj)
            s.proj = packing.conv_w(pk, self.projection.conv_skip_proj.weight, self.cin, scale=sc, bias=bi)
        s.downsample = int(bool(self._downsample))
        return pk

    def forward_nhwc(self, x):
        n, h, w, c = x.shape
        L = _lib.lib()
        ho, wo = ((h + 1) // 2, (w + 1) // 2) if self._downsample else (h, w)
        ws = runtime.workspace(L.sf_bottleneck_ws_bytes(self.cin, self.cout, n, h, w), x.device)
        out = torch.empty((n, ho, wo, self.cout), dtype=torch.float32, device=x.device)
        _lib.check(L.sf_bottleneck_fwd(self.packed().struct, ptr(x), ptr(out), n, h, w, ptr(ws), ws.numel() * 4,
                                       runtime.stream_ptr(x.device)), "bottleneck")
        return out

    def forward(self, *args):
        (x,) = args
        return runtime.to_nchw(self.forward_nhwc(runtime.to_nhwc(x)))


class LayerNorm(nn.Module):
    """Parameter container for convolutions.py:283-308; always applied fused into a conv epilogue."""

    def __init__(self, normalized_shape, eps=1e-6, data_format="channels_last"):
        super().__init__()
        if data_format not in ("channels_last", "channels_first"):
            raise NotImplementedError
        self.weight = nn.Parameter(torch.ones(normalized_shape))
        self.bias = nn.Parameter(torch.zeros(normalized_shape))
        self.eps, self.data_format, self.normalized_shape = eps, data_format, (normalized_shape,)


class Block(PackedModule):
    """ConvNeXt block: 7x7 depthwise -> LayerNorm -> Linear(4x) -> GELU -> Linear -> gamma -> residual."""

    def __init__(self, dim, drop_path=0., layer_scale_init_value=1e-6):
        super().__init__()
        # drop_path: stochastic depth is the identity at inference (timm's DropPath returns its input when not training);
        # accepted for signature compatibility — BiGRU passes Block(C, C) — and otherwise unused
        self.drop_path_rate = drop_path
        self.dwconv = nn.Conv2d(dim, dim, kernel_size=7, padding=3, groups=dim)
        self.norm = LayerNorm(dim, eps=1e-6)
        self.pwconv1 = nn.Linear(dim, 4 * dim)
        self.pwconv2 = nn.Linear(4 * dim, dim)
        self.gamma = nn.Parameter(layer_scale_init_value * torch.ones(dim)) if layer_scale_init_value > 0 else None
        self.dim = dim

    def _pack(self):
        pk = packing.Pack(_lib.ConvNextW())
        s, d = pk.struct, self.dim
        s.dw_w = pk.hold(self.dwconv.weight.reshape(d, 49).t())          # [49][C]
        s.dw_b = pk.hold(self.dwconv.bias)
        s.ln_w, s.ln_b = pk.hold(self.norm.weight), pk.hold(self.norm.bias)
        s.pw1 = packing.conv_w(pk, self.pwconv1.weight[:, :, None, None], d, bias=self.pwconv1.bias, act="gelu")
        gamma = self.gamma if self.gamma is not None else torch.ones_like(self.pwconv2.bias)
        s.pw2 = packing.conv_w(pk, self.pwconv2.weight[:, :, None, None], 4 * d, scale=gamma,
                               bias=gamma * self.pwconv2.bias)
        s.C = d
        return pk

    def forward_nhwc(self, x):
        n, h, w, c = x.shape
        L = _lib.lib()
        ws = runtime.workspace(L.sf_convnext_block_ws_bytes(c, n, h, w), x.device)
        out = torch.empty_like(x)
        _lib.check(L.sf_convnext_block_fwd(self.packed().struct, ptr(x), ptr(out), n, h, w, ptr(ws),
                                           ws.numel() * 4, runtime.stream_ptr(x.device)), "convnext_block")
        return out

    def forward(self, x):
        return runtime.to_nchw(self.forward_nhwc(runtime.to_nhwc(x)))


class Bottleblock(PackedModule):
    """7x7 + LN + GELU -> 1x1 + LN + GELU -> 3x3 + LN + GELU with a residual (the input itself, or a 1x1 + GELU projection
    of it when the channel count changes) — convolutions.py:348-380.  The trusting-gate body of the dual GRU cells (which
    run it fused with the gate's softmax / mix, layers/temporal_ode_bayes.py) and a block of ``BiGRU``; ``forward`` is the
    stand-alone form (``sf_bottleblock_fwd``: four launches, the residual added in the last epilogue)."""

    def __init__(self, in_channels, out_channels=None):
        super().__init__()
        mid = int(in_channels / 2)
        out_channels = out_channels or in_channels
        cf = dict(eps=1e-6, data_format="channels_first")
        self.layers = _seq(
            nn.Conv2d(in_channels, mid, kernel_size=7, bias=False, padding=3), LayerNorm(mid, **cf), nn.GELU(),
            nn.Conv2d(mid, mid, kernel_size=1, bias=False), LayerNorm(mid, **cf), nn.GELU(),
            nn.Conv2d(mid, out_channels, kernel_size=3, bias=False, padding=1), LayerNorm(out_channels, **cf), nn.GELU())
        self.projection = None if out_channels == in_channels else _seq(
            nn.Conv2d(in_channels, out_channels, kernel_size=1, bias=False), nn.GELU())
        self.in_channels, self.out_channels = in_channels, out_channels

    def _pack(self):
        pk = packing.Pack(_lib.BottleW())
        s, L, cin = pk.struct, self.layers, self.in_channels
        mid = L[0].out_channels
        s.c7 = packing.conv_w(pk, L[0].weight, cin, 0, scale=L[1].weight, bias=L[1].bias)
        s.c1 = packing.conv_w(pk, L[3].weight, mid, scale=L[4].weight, bias=L[4].bias)
        s.c3 = packing.conv_w(pk, L[6].weight, mid, scale=L[7].weight, bias=L[7].bias)
        if self.projection is not None:
            s.proj = packing.conv_w(pk, self.projection[0].weight, cin, 0)
        return pk

    def forward_nhwc(self, x0, x1=None):
        """x0 [n, h, w, c0] (and x1 [n, h, w, c1]: the block then reads cat[x0, x1] without materialising it)."""
        n, h, w, c0 = x0.shape
        c1 = 0 if x1 is None else x1.shape[-1]
        if c0 + c1 != self.in_channels:
            raise ValueError(f"Bottleblock({self.in_channels}) got {c0 + c1} input channels")
        st = self.packed().struct
        if c1:      # same packed weights, the channel split of this call
            if self.projection is None:
                raise ValueError("a Bottleblock without projection adds its input: pass one tensor")
            st = _lib.BottleW.from_buffer_copy(st)
            st.c7.c0, st.c7.c1, st.proj.c0, st.proj.c1 = c0, c1, c0, c1
        L = _lib.lib()
        ws = runtime.workspace(L.sf_bottleblock_ws_bytes(self.in_channels, self.out_channels, n, h, w), x0.device)
        out = torch.empty((n, h, w, self.out_channels), dtype=torch.float32, device=x0.device)
        _lib.check(L.sf_bottleblock_fwd(st, ptr(x0), ptr(x1), ptr(out), n, h, w, ptr(ws), ws.numel() * 4,
                                        runtime.stream_ptr(x0.device)), "bottleblock")
        return out

    def forward(self, *args):
        (x,) = args
        runtime.require_cuda(x)
        return runtime.to_nchw(self.forward_nhwc(runtime.to_nhwc(x)))


class ASPP(nn.Module):
    def __init__(self, in_channels, atrous_rates, out_channels=256):
        super().__init__()
        branches = [_seq(nn.Conv2d(in_channels, out_channels, 1, bias=False), nn.BatchNorm2d(out_channels), nn.ReLU())]
        for rate in tuple(atrous_rates):
            branches.append(_seq(nn.Conv2d(in_channels, out_channels, 3, padding=rate, dilation=rate, bias=False),
                                 nn.BatchNorm2d(out_channels), nn.ReLU()))
        branches.append(_seq(nn.AdaptiveAvgPool2d(1), nn.Conv2d(in_channels, out_channels, 1, bias=False),
                             nn.BatchNorm2d(out_channels), nn.ReLU()))
        self.convs = nn.ModuleList(branches)
        self.project = _seq(nn.Conv2d(len(self.convs) * out_channels, out_channels, 1, bias=False),
                            nn.BatchNorm2d(out_channels), nn.ReLU(), nn.Dropout(0.5))
        self.rates = tuple(atrous_rates)


class DeepLabHead(nn.Sequential, PackedModule):
    """ASPP(12,24,36) -> 3x3+BN+ReLU -> 1x1.  The image-pooling branch (global mean -> 1x1 -> BN ->
    ReLU -> constant broadcast) is folded into a per-image bias of the ASPP projection."""

    def __init__(self, in_channels, num_classes, hidden_channel=256):
        nn.Sequential.__init__(
            self, ASPP(in_channels, [12, 24, 36], hidden_channel),
            nn.Conv2d(hidden_channel, hidden_channel, 3, padding=1, bias=False),
            nn.BatchNorm2d(hidden_channel), nn.ReLU(), nn.Conv2d(hidden_channel, num_classes, 1))
        self.in_channels, self.hidden_channel = in_channels, hidden_channel

    def _pack(self):
        return self.pack_after(None)

    def pack_after(self, pre):
        """The packed head, optionally for an input that is still to go through a bias-free 1x1 convolution ``pre``
        ([C][C_in][1][1] weight): every consumer of the head's input is linear in it (the four branch convolutions, the mean of
        the pooling branch) and a bias-free 1x1 maps the zero padding to zeros, so W_branch . (pre . h) = (W_branch o pre) . h —
        the composition is formed once in fp64 and the 1x1 launches never run (FuturePredictionODE: SpatialGRU.conv_decoder)."""
        if self.training:
            raise RuntimeError("streamingflow_amd is inference-only: call .eval() (BatchNorm uses running statistics)")
        pk = packing.Pack(_lib.DeepLabW())
        s, aspp, C, hid = pk.struct, self[0], self.in_channels, self.hidden_channel
        if len(aspp.convs) != 5:
            raise NotImplementedError
        if pre is not None:
            assert pre.shape[0] == C and tuple(pre.shape[2:]) == (1, 1), tuple(pre.shape)
            C = pre.shape[1]
            pd = pre.detach()[:, :, 0, 0].double()
            fold = lambda w: torch.einsum("ockl,ci->oikl", w.detach().double(), pd).float()
        else:
            fold = lambda w: w
        for i in range(4):
            conv, bn = aspp.convs[i][0], aspp.convs[i][1]
            sc, bi = packing.bn_fold(bn)
            s.branch[i] = packing.conv_w(pk, fold(conv.weight), C, scale=sc, bias=bi, act="relu", dil=conv.dilation[0])
        sc, bi = packing.bn_fold(aspp.convs[4][2])
        s.pool_w = pk.hold(fold(aspp.convs[4][1].weight).reshape(hid, C))
        s.pool_scale, s.pool_bias = pk.hold(sc), pk.hold(bi)
        wproj = aspp.project[0].weight                                    # [hid][5*hid][1][1]
        s.proj_pool_w = pk.hold(wproj[:, 4 * hid:, 0, 0])
        sc, bi = packing.bn_fold(aspp.project[1])
        s.project = packing.conv_w(pk, wproj[:, :4 * hid], 4 * hid, scale=sc, bias=bi, act="relu")
        sc, bi = packing.bn_fold(self[2])
        s.conv3 = packing.conv_w(pk, self[1].weight, hid, scale=sc, bias=bi, act="relu")
        s.cls = packing.conv_w(pk, self[4].weight, hid, bias=self[4].bias)
        s.C, s.hid = C, hid
        return pk

    def forward_nhwc(self, x, st=None):
        n, h, w, c = x.shape
        L = _lib.lib()
        st = self.packed().struct if st is None else st
        ws = runtime.workspace(L.sf_deeplab_head_ws_bytes(c, self.hidden_channel, n, h, w), x.device)
        out = torch.empty((n, h, w, st.cls.cout), dtype=torch.float32, device=x.device)
        _lib.check(L.sf_deeplab_head_fwd(st, ptr(x), ptr(out), n, h, w, ptr(ws), ws.numel() * 4,
                                         runtime.stream_ptr(x.device)), "deeplab_head")
        return out

    def forward_nhwc_into_planar(self, x, out, group, stride_major, stride_minor, st=None):
        """The same head, its classifier writing the reference's planar layout itself: image i of x [n, h, w, C] lands as
        [cout][h][w] planes at ``out.data_ptr() + 4 * ((i // group) * stride_major + (i % group) * stride_minor)`` (strides in
        floats) — FuturePredictionODE writes frames (t, b) straight into its [B, T, C, H, W] result, no transpose launches."""
        n, h, w, c = x.shape
        L = _lib.lib()
        st = self.packed().struct if st is None else st
        ws = runtime.workspace(L.sf_deeplab_head_ws_bytes(c, self.hidden_channel, n, h, w), x.device)
        _lib.check(L.sf_deeplab_head_planar_fwd(st, ptr(x), ptr(out), n, h, w, int(group), int(stride_major), int(stride_minor),
                                                ptr(ws), ws.numel() * 4, runtime.stream_ptr(x.device)), "deeplab_head_planar")
        return out

    def forward(self, x):
        return runtime.to_nchw(self.forward_nhwc(runtime.to_nhwc(x)))
