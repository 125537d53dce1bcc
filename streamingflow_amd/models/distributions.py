"""MI355X-native counterpart of streamingflow/models/distributions.py (``DistributionModule``
:7-51, ``DistributionEncoder`` :54-68).  The reference imports but never instantiates it
(SURVEY.md §0); kept for signature / state_dict compatibility: all three methods."""
import ctypes

import torch
import torch.nn as nn

from .. import _lib, packing, runtime
from ..layers.convolutions import Bottleneck
from ..runtime import PackedModule, ptr


class DistributionEncoder(nn.Module):
    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.model = nn.Sequential(Bottleneck(in_channels, out_channels=out_channels, downsample=True),
                                   Bottleneck(out_channels, out_channels=out_channels, downsample=True),
                                   Bottleneck(out_channels, out_channels=out_channels, downsample=True),
                                   Bottleneck(out_channels, out_channels=out_channels, downsample=True))

    def forward_nhwc(self, x):
        for blk in self.model:
            x = blk.forward_nhwc(x)
        return x

    def forward(self, s_t):
        return runtime.to_nchw(self.forward_nhwc(runtime.to_nhwc(s_t)))


class DistributionModule(PackedModule):
    """``method``: 'GAUSSIAN' (2 * latent_dim parameters), 'MIXGAUSSIAN' (6 * latent_dim + 3) — both encoder + global average pool +
    1x1 conv — or 'BERNOULLI' (one Bottleneck to latent_dim channels + LogSigmoid, a [b, latent_dim, H, W] map)."""

    def __init__(self, in_channels, latent_dim, method="GAUSSIAN"):
        super().__init__()
        self.compress_dim, self.latent_dim, self.method = in_channels // 2, latent_dim, method
        if method in ("GAUSSIAN", "MIXGAUSSIAN"):
            self.n_out = 2 * latent_dim if method == "GAUSSIAN" else 6 * latent_dim + 3
            self.encoder = DistributionEncoder(in_channels, self.compress_dim)
            self.decoder = nn.Sequential(nn.AdaptiveAvgPool2d(1), nn.Conv2d(self.compress_dim, out_channels=self.n_out, kernel_size=1))
        elif method == "BERNOULLI":
            self.encoder = nn.Sequential(Bottleneck(in_channels, latent_dim))
            self.decoder = nn.LogSigmoid()
        else:
            raise NotImplementedError

    def _pack(self):
        pk = packing.Pack(None)
        if self.method != "BERNOULLI":
            conv = self.decoder[1]
            w, b = conv.weight, conv.bias
            pad = (-self.n_out) % 4                                # the kernels move output channels in fours (6 * latent + 3 is odd)
            if pad:
                w = torch.cat([w, w.new_zeros((pad,) + tuple(w.shape[1:]))], 0)
                b = torch.cat([b, b.new_zeros(pad)], 0)
            pk.struct = packing.conv_w(pk, w, self.compress_dim, bias=b)
        return pk

    def forward(self, s_t):
        b, s = s_t.shape[:2]
        assert s == 1
        runtime.require_cuda(s_t)
        L = _lib.lib()
        if self.method == "BERNOULLI":
            enc = self.encoder[0].forward_nhwc(runtime.to_nhwc(s_t[:, 0]))
            out = torch.empty_like(enc)
            _lib.check(L.sf_logsigmoid_fwd(ptr(enc), ptr(out), enc.numel(), runtime.stream_ptr(enc.device)), "logsigmoid")
            return runtime.to_nchw(out)
        enc = self.encoder.forward_nhwc(runtime.to_nhwc(s_t[:, 0]))
        n, h, w, c = enc.shape
        st = self.packed().struct
        ws = runtime.workspace(L.sf_dist_head_ws_bytes(c, n), enc.device)
        out = torch.empty((n, 1, 1, st.cout), dtype=torch.float32, device=enc.device)
        _lib.check(L.sf_dist_head_fwd(ctypes.byref(st), ptr(enc), ptr(out), n, h, w, 1, 0, 0.0, 0.0,
                                      ptr(ws), ws.numel() * 4, runtime.stream_ptr(enc.device)), "dist_head")
        return out.view(b, 1, st.cout)[:, :, : self.n_out]
