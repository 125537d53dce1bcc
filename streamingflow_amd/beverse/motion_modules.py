"""MI355X-native counterparts of mmdet3d/models/beverse/models/motion_modules.py:
``DistributionModule`` (:10-46), ``SpatialDistributionModule`` (:49-88), ``DistributionEncoder``
(:91-108) and ``FuturePrediction`` (:111-146).  Same signatures and state_dict keys; HIP execution."""
import torch
import torch.nn as nn

from .. import _lib, packing, runtime
from ..runtime import PackedModule, ptr
from .basic_modules import Bottleneck, SpatialGRU


class DistributionEncoder(nn.Module):
    def __init__(self, in_channels, out_channels, num_layer=2):
        super().__init__()
        layers = []
        for _ in range(num_layer):
            layers.append(Bottleneck(in_channels=in_channels, out_channels=out_channels, downsample=True))
            in_channels = out_channels
        self.model = nn.Sequential(*layers)

    def forward_nhwc(self, x):
        for blk in self.model:
            x = blk.forward_nhwc(x)
        return x

    def forward(self, s_t):
        return runtime.to_nchw(self.forward_nhwc(runtime.to_nhwc(s_t)))


class _DistBase(PackedModule):
    global_pool = False

    def __init__(self, in_channels, latent_dim, min_log_sigma, max_log_sigma):
        super().__init__()
        self.compress_dim, self.latent_dim = in_channels // 2, latent_dim
        self.min_log_sigma, self.max_log_sigma = min_log_sigma, max_log_sigma
        self.encoder = DistributionEncoder(in_channels, self.compress_dim)
        head = nn.Conv2d(self.compress_dim, out_channels=2 * latent_dim, kernel_size=1)
        self.last_conv = nn.Sequential(nn.AdaptiveAvgPool2d(1), head) if self.global_pool else nn.Sequential(head)

    def _pack(self):
        pk = packing.Pack(None)
        conv = self.last_conv[-1]
        pk.struct = packing.conv_w(pk, conv.weight, self.compress_dim, bias=conv.bias)
        return pk

    def _head(self, enc):
        n, h, w, c = enc.shape
        L = _lib.lib()
        ws = runtime.workspace(L.sf_dist_head_ws_bytes(c, n), enc.device)
        ho, wo = (1, 1) if self.global_pool else (h, w)
        out = torch.empty((n, ho, wo, 2 * self.latent_dim), dtype=torch.float32, device=enc.device)
        import ctypes
        _lib.check(L.sf_dist_head_fwd(ctypes.byref(self.packed().struct), ptr(enc), ptr(out), n, h, w,
                                      int(self.global_pool), 1, float(self.min_log_sigma), float(self.max_log_sigma),
                                      ptr(ws), ws.numel() * 4, runtime.stream_ptr(enc.device)), "dist_head")
        return out


class DistributionModule(_DistBase):
    """Diagonal Gaussian over the whole BEV map: (mu, log_sigma) of shape (b, 1, latent_dim)."""
    global_pool = True

    def forward(self, s_t):
        b, s = s_t.shape[:2]
        assert s == 1
        runtime.require_cuda(s_t)
        out = self._head(self.encoder.forward_nhwc(runtime.to_nhwc(s_t[:, 0]))).view(b, 1, 2 * self.latent_dim)
        return out[:, :, :self.latent_dim], out[:, :, self.latent_dim:]


class SpatialDistributionModule(_DistBase):
    """Per-location Gaussian (no global pooling): (mu, log_sigma) of shape (b, latent_dim, h/4, w/4)."""
    global_pool = False

    def forward(self, s_t):
        b, s = s_t.shape[:2]
        assert s == 1
        runtime.require_cuda(s_t)
        out = runtime.to_nchw(self._head(self.encoder.forward_nhwc(runtime.to_nhwc(s_t[:, 0]))))
        return out[:, :self.latent_dim], out[:, self.latent_dim:]


class FuturePrediction(nn.Module):
    """[SpatialGRU -> n_res_layers x Bottleneck] x n_gru_blocks (motion_modules.py:111-146)."""

    def __init__(self, in_channels, latent_dim, n_gru_blocks=3, n_res_layers=3):
        super().__init__()
        self.n_gru_blocks = n_gru_blocks
        grus, blocks = [], []
        for i in range(n_gru_blocks):
            grus.append(SpatialGRU(latent_dim if i == 0 else in_channels, in_channels))
            blocks.append(nn.Sequential(*[Bottleneck(in_channels) for _ in range(n_res_layers)]))
        self.spatial_grus = nn.ModuleList(grus)
        self.res_blocks = nn.ModuleList(blocks)

    def forward(self, x, hidden_state):
        runtime.require_cuda(x, hidden_state)
        b, T, c, h, w = x.shape
        xn = runtime.to_nhwc(x.reshape(b * T, c, h, w)).view(b, T, h, w, c).permute(1, 0, 2, 3, 4).contiguous()
        hid = runtime.to_nhwc(hidden_state)
        for gru, blocks in zip(self.spatial_grus, self.res_blocks):
            xn = gru.forward_nhwc(xn, hid)                       # [T, b, h, w, C]
            y = xn.view(T * b, h, w, xn.shape[-1])
            for blk in blocks:
                y = blk.forward_nhwc(y)
            xn = y.view(T, b, h, w, y.shape[-1])
        C = xn.shape[-1]
        out = runtime.to_nchw(xn.permute(1, 0, 2, 3, 4).reshape(b * T, h, w, C))
        return out.view(b, T, C, h, w)
