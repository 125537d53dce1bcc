"""MI355X-native ``SpatialGRU`` (streamingflow/layers/temporal.py:11-57): a conv-GRU run over the
T frames of a [B, T, C, H, W] tensor followed by a 1x1 decoder, on libsfnative (HIP, gfx950) — and the two
recurrent modules the reference defines beside it but never constructs, ``Dual_GRU`` (:59-152) and ``BiGRU``
(:154-249), composed from the same fused cells.

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

    def pack_states_only(self):
        """The cell without its decoder: the library then returns the hidden states themselves (a consumer that folds
        ``conv_decoder`` into its own weights reads them: DeepLabHead.pack_after)."""
        pk = packing.Pack(None)
        pk.struct = pack_gru(pk, self.conv_update, self.conv_reset, self.conv_state_tilde, self.input_size, self.hidden_size, None,
                             gate_bias=self.gru_bias_init)
        return pk

    def forward_nhwc(self, x, state, st=None):
        """x: [T, B, H, W, Cx] (or [T, H, W, Cx] for one sample), state: [B, H, W, C] (or [H, W, C]).
        Returns the decoded outputs with the shape of x (the hidden states [.., C] with the struct of ``pack_states_only``)."""
        one = x.dim() == 4
        if one:
            x, state = x[:, None], state[None]
        T, B, h, w, _ = x.shape
        L = _lib.lib()
        ws = runtime.workspace(L.sf_spatial_gru_ws_bytes(self.hidden_size, B, h, w), x.device)
        out = torch.empty((T, B, h, w, self.input_size if st is None else self.hidden_size), dtype=torch.float32, device=x.device)
        _lib.check(L.sf_spatial_gru_fwd(self.packed().struct if st is None else st, ptr(x), ptr(state), ptr(out), T, B, h, w, ptr(ws),
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


# ---- building blocks on NHWC tensors (shared with the dual cells of temporal_ode_bayes.py) ---------------------------

def gru_cell_nhwc(gw, x, s):
    """One conv-GRU update (1 - u) * s + u * h~ (temporal.py:126-152): x [n, h, w, Cx], s [n, h, w, C]."""
    n, h, w, C = s.shape
    L = _lib.lib()
    ws = runtime.workspace(L.sf_gru_cell_ws_bytes(C, n, h, w), s.device)
    out = torch.empty_like(s)
    _lib.check(L.sf_gru_cell_fwd(gw, ptr(x), ptr(s), ptr(out), n, h, w, ptr(ws), ws.numel() * 4, runtime.stream_ptr(s.device)),
               "gru_cell")
    return out


def conv_nhwc(cw, x):
    n, h, w, _ = x.shape
    out = torch.empty((n, h, w, cw.cout), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().sf_conv2d_fwd(cw, ptr(x), None, None, ptr(out), n, h, w, 0, runtime.stream_ptr(x.device)), "conv2d")
    return out


def trust_mix_nhwc(dw, r1, r2):
    """softmax(trusting_gate(cat[r1, r2])) mixes the two branch states: r2 * g0 + r1 * g1 (temporal.py:118-122)."""
    n, h, w, C = r1.shape
    L = _lib.lib()
    ws = runtime.workspace(L.sf_dual_cell_ws_bytes(C, n, h, w), r1.device)
    out = torch.empty_like(r1)
    _lib.check(L.sf_trust_mix_fwd(dw, ptr(r1), ptr(r2), None, ptr(out), 0, None, None, n, h, w, ptr(ws), ws.numel() * 4,
                                  runtime.stream_ptr(r1.device)), "trust_mix")
    return out


def pack_trust_gate(pk, dw, gate, C):
    """Fill the trusting-gate fields of a ``DualW`` from ``nn.Sequential(Bottleblock(2C, C), Conv2d(C, 2, 1))``."""
    bb = gate[0]
    L = bb.layers
    dw.tg7 = packing.conv_w(pk, L[0].weight, C, C, scale=L[1].weight, bias=L[1].bias)
    dw.tg1 = packing.conv_w(pk, L[3].weight, C, scale=L[4].weight, bias=L[4].bias)
    dw.tg3 = packing.conv_w(pk, L[6].weight, C, scale=L[7].weight, bias=L[7].bias)
    dw.tgproj = packing.conv_w(pk, bb.projection[0].weight, C, C)
    dw.w_logit = pk.hold(gate[1].weight.reshape(2, C))
    dw.C = C


class Dual_GRU(PackedModule):
    """Two conv-GRU branches rolled out for ``n_future`` steps, mixed per step by the trusting gate
    (temporal.py:59-152; defined by the reference, constructed nowhere).  forward(x [b, 1, Cin, h, w],
    state [b, n_present, C, h, w]) -> [b, n_future, C, h, w].  Per step: two fused GRU cells, the 3x3 decoder of
    branch 2 and the fused gate (4 launches), on all b samples at once."""

    def __init__(self, in_channels, latent_dim, n_future, mixture=True, gru_bias_init=0.0):
        super().__init__()
        from .convolutions import Bottleblock
        if latent_dim % 8 or latent_dim > 128 or in_channels % 4:
            raise NotImplementedError("Dual_GRU: latent_dim a multiple of 8, <= 128 (the gate's LayerNorm layers keep a pixel's channels in one wave)")
        self.n_future, self.mixture = n_future, mixture
        self.input_size, self.hidden_size, self.gru_bias_init = in_channels, latent_dim, gru_bias_init
        for tag, cx in (("1", in_channels), ("2", latent_dim)):
            for name in ("conv_update_", "conv_reset_", "conv_state_tilde_"):
                setattr(self, name + tag, nn.Conv2d(cx + latent_dim, latent_dim, kernel_size=3, bias=True, padding=1))
        self.conv_decoder_2 = nn.Conv2d(latent_dim, latent_dim, kernel_size=3, bias=True, padding=1)
        self.trusting_gate = nn.Sequential(Bottleblock(2 * latent_dim, latent_dim), nn.Conv2d(latent_dim, 2, kernel_size=1, bias=False))

    def _pack(self):
        pk = packing.Pack(None)
        C, gb = self.hidden_size, self.gru_bias_init
        g1 = pack_gru(pk, self.conv_update_1, self.conv_reset_1, self.conv_state_tilde_1, self.input_size, C, gate_bias=gb)
        g2 = pack_gru(pk, self.conv_update_2, self.conv_reset_2, self.conv_state_tilde_2, C, C, gate_bias=gb)
        dec2 = packing.conv_w(pk, self.conv_decoder_2.weight, C, bias=self.conv_decoder_2.bias)
        dw = _lib.DualW()
        pack_trust_gate(pk, dw, self.trusting_gate, C)
        pk.struct = (g1, g2, dec2, dw)
        return pk

    def forward(self, x, state):
        runtime.require_cuda(x, state)
        b, s, c, h, w = x.shape
        assert c == self.input_size, f'feature sizes must match, got input {c} for layer with size {self.input_size}'
        g1, g2, dec2, dw = self.packed().struct
        n_present = state.shape[1]
        frames = [runtime.to_nhwc(state[:, t]) for t in range(n_present)]
        xn = runtime.to_nhwc(x[:, 0])
        hid = frames[0]
        for t in range(n_present - 1):                       # warm-up of branch 2 over the present frames
            hid = gru_cell_nhwc(g2, frames[t], hid)
        r1 = r2 = frames[-1]
        preds = []
        for _ in range(self.n_future):
            r1 = gru_cell_nhwc(g1, xn, r1)
            hid = gru_cell_nhwc(g2, r2, hid)
            r2 = conv_nhwc(dec2, hid)
            cur = trust_mix_nhwc(dw, r1, r2)
            preds.append(cur)
            if self.mixture:
                r1 = r2 = cur
        return torch.stack([runtime.to_nchw(p) for p in preds], dim=1)


class BiGRU(PackedModule):
    """A forward and a backward conv-GRU over the frames of x [b, s, C, h, w], each state decoded by a Bottleblock, the two
    sequences concatenated on channels and refined by Bottleblock(2C -> C) + two ConvNeXt blocks (temporal.py:154-249;
    defined by the reference, constructed nowhere).  The Bottleblock after the concat reads the two decoded sequences as
    a two-tensor input: the [b, s, 2C, h, w] concat is never written."""

    def __init__(self, in_channels, gru_bias_init=0.0):
        super().__init__()
        from .convolutions import Block, Bottleblock
        if in_channels % 8 or in_channels > 128:
            raise NotImplementedError("BiGRU: in_channels a multiple of 8, <= 128")
        self.input_size = self.hidden_size = in_channels
        self.gru_bias_init = gru_bias_init
        for tag in ("1", "2"):
            for name in ("conv_update_", "conv_reset_", "conv_state_tilde_"):
                setattr(self, name + tag, nn.Conv2d(2 * in_channels, in_channels, kernel_size=3, bias=True, padding=1))
            setattr(self, "conv_decoder_" + tag, Bottleblock(in_channels, in_channels))
        self.res_blocks = nn.Sequential(Bottleblock(in_channels + in_channels, in_channels), Block(in_channels, in_channels),
                                        Block(in_channels, in_channels))

    def _pack(self):
        pk = packing.Pack(None)
        C, gb = self.hidden_size, self.gru_bias_init
        pk.struct = (pack_gru(pk, self.conv_update_1, self.conv_reset_1, self.conv_state_tilde_1, C, C, gate_bias=gb),
                     pack_gru(pk, self.conv_update_2, self.conv_reset_2, self.conv_state_tilde_2, C, C, gate_bias=gb))
        return pk

    def forward(self, x):
        runtime.require_cuda(x)
        b, s, c, h, w = x.shape
        g1, g2 = self.packed().struct
        frames = [runtime.to_nhwc(x[:, t]) for t in range(s)]
        r1, r2 = frames[0], frames[-1]
        fwd, bwd = [], []
        for t in range(s):
            r1 = gru_cell_nhwc(g1, frames[t], r1)
            r2 = gru_cell_nhwc(g2, frames[s - t - 1], r2)
            fwd.append(self.conv_decoder_1.forward_nhwc(r1))
            bwd.append(self.conv_decoder_2.forward_nhwc(r2))
        a = torch.stack(fwd, dim=1).view(b * s, h, w, c)                 # frame-major per sample, as states.view(b * s, ...)
        z = torch.stack(bwd[::-1], dim=1).view(b * s, h, w, c)
        y = self.res_blocks[0].forward_nhwc(a, z)
        y = self.res_blocks[2].forward_nhwc(self.res_blocks[1].forward_nhwc(y))
        return runtime.to_nchw(y).view(b, s, c, h, w)
