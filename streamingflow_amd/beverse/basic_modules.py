"""MI355X-native counterparts of mmdet3d/models/beverse/models/basic_modules.py: ``ConvBlock``
(:11-64, parameter container), ``Bottleneck`` (:68-178) and ``SpatialGRU`` (:225-284 — conv-GRU whose
candidate is conv+BN+ReLU and whose output is the state sequence itself, no 1x1 decoder)."""
import torch
import torch.nn as nn

from .. import _lib, packing, runtime
from ..layers.convolutions import Bottleneck  # noqa: F401  (same block, same state_dict keys)
from ..runtime import PackedModule, ptr


class ConvBlock(nn.Module):
    def __init__(self, in_channels, out_channels=None, kernel_size=3, stride=1, norm='bn', activation='relu',
                 bias=False, transpose=False):
        super().__init__()
        if transpose or norm != 'bn' or activation != 'relu' or stride != 1:
            raise NotImplementedError("BEVerse ConvBlock: conv + BatchNorm + ReLU form only")
        out_channels = out_channels or in_channels
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding=int((kernel_size - 1) / 2), bias=bias)
        self.norm = nn.BatchNorm2d(out_channels)
        self.activation = nn.ReLU(inplace=True)


class SpatialGRU(PackedModule):
    def __init__(self, input_size, hidden_size, gru_bias_init=0.0, norm='bn', activation='relu'):
        super().__init__()
        self.input_size, self.hidden_size, self.gru_bias_init = input_size, hidden_size, gru_bias_init
        cat = input_size + hidden_size
        self.conv_update = nn.Conv2d(cat, hidden_size, kernel_size=3, bias=True, padding=1)
        self.conv_reset = nn.Conv2d(cat, hidden_size, kernel_size=3, bias=True, padding=1)
        self.conv_state_tilde = ConvBlock(cat, hidden_size, kernel_size=3, bias=False, norm=norm, activation=activation)

    def _pack(self):
        if self.training:
            raise RuntimeError("streamingflow_amd is inference-only: call .eval()")
        pk = packing.Pack(_lib.GruW())
        s = pk.struct
        wg = torch.cat([self.conv_update.weight, self.conv_reset.weight], 0)
        bg = torch.cat([self.conv_update.bias, self.conv_reset.bias], 0) + float(self.gru_bias_init)
        s.gates = packing.conv_w(pk, wg, self.input_size, self.hidden_size, bias=bg, act="sigmoid")
        sc, bi = packing.bn_fold(self.conv_state_tilde.norm)
        s.cand = packing.conv_w(pk, self.conv_state_tilde.conv.weight, self.input_size, self.hidden_size, scale=sc,
                                bias=bi, act="relu")
        return pk

    def forward_nhwc(self, x, state):
        """x [T, B, h, w, Cx], state [B, h, w, C] -> states [T, B, h, w, C]."""
        T, B, h, w, _ = x.shape
        L = _lib.lib()
        ws = runtime.workspace(L.sf_spatial_gru_ws_bytes(self.hidden_size, B, h, w), x.device)
        out = torch.empty((T, B, h, w, self.hidden_size), dtype=torch.float32, device=x.device)
        _lib.check(L.sf_spatial_gru_fwd(self.packed().struct, ptr(x), ptr(state), ptr(out), T, B, h, w, ptr(ws),
                                        ws.numel() * 4, runtime.stream_ptr(x.device)), "beverse_spatial_gru")
        return out

    def forward(self, x, state=None, flow=None, mode='bilinear'):
        assert len(x.size()) == 5, 'Input tensor must be BxTxCxHxW.'
        if flow is not None:
            raise NotImplementedError("flow warping is not on the StreamingFlow path")
        runtime.require_cuda(x, state)
        b, T, c, h, w = x.size()
        assert c == self.input_size, f'feature sizes must match, got input {c} for layer with size {self.input_size}'
        xn = runtime.to_nhwc(x.reshape(b * T, c, h, w)).view(b, T, h, w, c).permute(1, 0, 2, 3, 4).contiguous()
        s0 = (torch.zeros((b, h, w, self.hidden_size), dtype=torch.float32, device=x.device) if state is None
              else runtime.to_nhwc(state))
        out = self.forward_nhwc(xn, s0).permute(1, 0, 2, 3, 4).reshape(b * T, h, w, self.hidden_size)
        return runtime.to_nchw(out).view(b, T, self.hidden_size, h, w)
