"""MI355X-native ``SpatialGRU`` (streamingflow/layers/temporal.py:11-57): a conv-GRU run over the
T frames of a [B, T, C, H, W] tensor followed by a 1x1 decoder, on libsfnative (HIP, gfx950).

Per frame two fused kernels: (1) both gates as one 3x3 implicit GEMM with sigmoid epilogue,
(2) the candidate 3x3 GEMM reading cat[x, (1-r)*s] with the reset multiply applied while
staging and the blend (1-u)*s + u*h as its epilogue; then the 1x1 decoder.  Inference only.
"""
import torch
import torch.nn as nn

from .. import _lib, packing, runtime
from ..runtime import PackedModule, ptr


def pack_gru(pk, update, reset, tilde, cx, ch, decoder=None, fold_dup=False, gate_bias=0.0):
    """Pack a conv-GRU cell: gates = [update; reset] stacked on the output-channel axis.  ``gate_bias`` is the
    reference's ``gru_bias_init`` — a constant added to both gate pre-activations (temporal.py:50-51) — folded into the
    gates' bias."""
    s = _lib.GruW()
    wg = torch.cat([update.weight, reset.weight], 0)
    bg = torch.cat([update.bias, reset.bias], 0) + float(gate_bias)
    if fold_dup:
        s.gates = packing.conv_w(pk, wg, ch, 0, bias=bg, act="sigmoid", fold_dup=True)
    else:
        s.gates = packing.conv_w(pk, wg, cx, ch, bias=bg, act="sigmoid")
    s.cand = packing.conv_w(pk, tilde.weight, cx, ch, bias=tilde.bias)
    if decoder is not None:
        s.decoder = packing.conv_w(pk, decoder.weight, ch, bias=decoder.bias)
    return s


class SpatialGRU(PackedModule):
    def __init__(self, input_size, hidden_size, gru_bias_init=0.0):
        super().__init__()
        self.input_size, self.hidden_size, self.gru_bias_init = input_size, hidden_size, gru_bias_init
        cat = input_size + hidden_size
        self.conv_update = nn.Conv2d(cat, hidden_size, kernel_size=3, bias=True, padding=1)
        self.conv_reset = nn.Conv2d(cat, hidden_size, kernel_size=3, bias=True, padding=1)
        self.conv_state_tilde = nn.Conv2d(cat, hidden_size, kernel_size=3, bias=True, padding=1)
        self.conv_decoder = nn.Conv2d(hidden_size, input_size, kernel_size=1, bias=False)

    def _pack(self):
        pk = packing.Pack(None)
        pk.struct = pack_gru(pk, self.conv_update, self.conv_reset, self.conv_state_tilde, self.input_size,
                             self.hidden_size, self.conv_decoder, gate_bias=self.gru_bias_init)
        return pk

    def forward_nhwc(self, x, state):
        """x: [T, B, H, W, Cx] (or [T, H, W, Cx] for one sample), state: [B, H, W, C] (or [H, W, C]).
        Returns the decoded outputs with the shape of x."""
        one = x.dim() == 4
        if one:
            x, state = x[:, None], state[None]
        T, B, h, w, _ = x.shape
        L = _lib.lib()
        ws = runtime.workspace(L.sf_spatial_gru_ws_bytes(self.hidden_size, B, h, w), x.device)
        out = torch.empty((T, B, h, w, self.input_size), dtype=torch.float32, device=x.device)
        _lib.check(L.sf_spatial_gru_fwd(self.packed().struct, ptr(x), ptr(state), ptr(out), T, B, h, w, ptr(ws),
                                        ws.numel() * 4, runtime.stream_ptr(x.device)), "spatial_gru")
        return out[:, 0] if one else out

    def gru_cell(self, x, state):
        """One cell update on NCHW tensors (temporal.py:44-57)."""
        runtime.require_cuda(x, state)
        xn, sn = runtime.to_nhwc(x), runtime.to_nhwc(state)
        n, h, w, _ = xn.shape
        L = _lib.lib()
        ws = runtime.workspace(L.sf_gru_cell_ws_bytes(self.hidden_size, n, h, w), x.device)
        out = torch.empty_like(sn)
        _lib.check(L.sf_gru_cell_fwd(self.packed().struct, ptr(xn), ptr(sn), ptr(out), n, h, w, ptr(ws), ws.numel() * 4,
                                     runtime.stream_ptr(x.device)), "gru_cell")
        return runtime.to_nchw(out)

    def forward(self, x, state=None):
        assert len(x.size()) == 5, 'Input tensor must be BxTxCxHxW.'
        runtime.require_cuda(x, state)
        b, T, c, h, w = x.shape
        xn = runtime.to_nhwc(x.reshape(b * T, c, h, w)).view(b, T, h, w, c).permute(1, 0, 2, 3, 4).contiguous()
        s0 = (torch.zeros((b, h, w, self.hidden_size), dtype=torch.float32, device=x.device) if state is None
              else runtime.to_nhwc(state))
        out = self.forward_nhwc(xn, s0).permute(1, 0, 2, 3, 4).reshape(b * T, h, w, self.input_size)
        return runtime.to_nchw(out).view(b, T, self.input_size, h, w)
