"""MI355X-native counterpart of streamingflow/models/distributions.py (``DistributionModule``
:7-51, ``DistributionEncoder`` :54-68).  The reference imports but never instantiates it
(SURVEY.md §0); kept for signature / state_dict compatibility.  GAUSSIAN method only."""
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
    def __init__(self, in_channels, latent_dim, method="GAUSSIAN"):
        super().__init__()
        if method != "GAUSSIAN":
            raise NotImplementedError("only method='GAUSSIAN' is built (MIXGAUSSIAN / BERNOULLI are unused by the reference)")
        self.compress_dim, self.latent_dim, self.method = in_channels // 2, latent_dim, method
        self.encoder = DistributionEncoder(in_channels, self.compress_dim)
        self.decoder = nn.Sequential(nn.AdaptiveAvgPool2d(1),
                                     nn.Conv2d(self.compress_dim, out_channels=2 * latent_dim, kernel_size=1))

    def _pack(self):
        pk = packing.Pack(None)
        conv = self.decoder[1]
        pk.struct = packing.conv_w(pk, conv.weight, self.compress_dim, bias=conv.bias)
        return pk

    def forward(self, s_t):
        b, s = s_t.shape[:2]
        assert s == 1
        runtime.require_cuda(s_t)
        enc = self.encoder.forward_nhwc(runtime.to_nhwc(s_t[:, 0]))
        n, h, w, c = enc.shape
        L = _lib.lib()
        ws = runtime.workspace(L.sf_dist_head_ws_bytes(c, n), enc.device)
        out = torch.empty((n, 1, 1, 2 * self.latent_dim), dtype=torch.float32, device=enc.device)
        _lib.check(L.sf_dist_head_fwd(ctypes.byref(self.packed().struct), ptr(enc), ptr(out), n, h, w, 1, 0, 0.0, 0.0,
                                      ptr(ws), ws.numel() * 4, runtime.stream_ptr(enc.device)), "dist_head")
        return out.view(b, 1, 2 * self.latent_dim)
