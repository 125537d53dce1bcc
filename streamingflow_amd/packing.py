"""Load-time weight packing: reference-format parameters (OIHW fp32, BatchNorm buffers, ...) ->
the packed device layout of include/sfnative.h.  Runs once per (module, parameter version); uses
torch tensor ops on the parameters' own device as plumbing — nothing here is on the hot path.
"""
import torch

from . import _lib

def _round_up(v, m):
    return (v + m - 1) // m * m


class Pack:
    """Owns the packed tensors (keeps them alive) next to the ctypes struct that points at them."""

    def __init__(self, struct):
        self.struct = struct
        self.keep = []

    def hold(self, t):
        if t is None:
            return None
        t = t.detach().to(torch.float32).contiguous()
        self.keep.append(t)
        return t.data_ptr()

    def adopt(self, other):
        self.keep.append(other)
        return other.struct


def bn_fold(bn, conv_bias=None):
    """eval-mode BatchNorm2d after a conv -> per-channel (scale, bias) on the accumulator."""
    sc = bn.weight.detach() / torch.sqrt(bn.running_var.detach() + bn.eps)
    bi = bn.bias.detach() - bn.running_mean.detach() * sc
    if conv_bias is not None:
        bi = bi + conv_bias.detach() * sc
    return sc, bi


def conv_w(holder, weight, c0, c1=0, scale=None, bias=None, act="none", dil=1, stride=1, pad=None,
           transposed=False, fold_dup=False, interleave=False):
    """Pack one convolution.  weight: Conv2d [cout][cin][kh][kw] (or ConvTranspose2d [cin][cout][kh][kw]
    with transposed=True, k3/s1/p1 only).  Returns a filled _lib.ConvW whose tensors `holder` owns."""
    w = weight.detach().to(torch.float32)
    if transposed:
        w = w.permute(1, 0, 2, 3).flip(2, 3)
    if fold_dup:   # layer reads cat[s, s]: W[:, :C] + W[:, C:] applied to s once
        half = w.shape[1] // 2
        w = w[:, :half] + w[:, half:]
    cout, cin, kh, kw = w.shape
    assert cin == c0 + c1, (cin, c0, c1)
    cin_pad, cout_pad = _round_up(cin, 32), _round_up(cout, 16)
    dev = w.device

    def pad_vec(v):
        if v is None:
            return None
        out = torch.zeros(cout_pad, device=dev, dtype=torch.float32)
        out[:cout] = v.detach().to(torch.float32)
        return out

    packed = torch.zeros(cout_pad, kh, kw, cin_pad, device=dev, dtype=torch.float32)
    packed[:cout, :, :, :cin] = w.permute(0, 2, 3, 1)
    scale, bias = pad_vec(scale), pad_vec(bias)
    if interleave:
        # row 16T+4g+r <- (r<2: loc channel 8T+2g+r) / (r>=2: raw channel C+8T+2g+r-2)
        Ch = cout // 2
        cout_pad = _round_up(_round_up(Ch, 8) * 2, 16)
        idx = torch.full((cout_pad,), -1, dtype=torch.long)
        for row in range(cout_pad):
            T, g, r = row // 16, (row % 16) // 4, row % 4
            c = 8 * T + 2 * g + (r & 1)
            if c < Ch:
                idx[row] = c if r < 2 else Ch + c
        sel = idx.clamp(min=0).to(dev)
        mask = (idx >= 0).to(dev)
        packed = packed[sel] * mask[:, None, None, None]
        if scale is not None:
            scale = scale[sel] * mask
        if bias is not None:
            bias = bias[sel] * mask
    s = _lib.ConvW()
    s.w = holder.hold(packed.reshape(cout_pad, kh * kw * cin_pad))
    s.scale = holder.hold(scale)
    s.bias = holder.hold(bias)
    s.cout, s.cout_pad, s.c0, s.c1, s.cin_pad = cout, cout_pad, c0, c1, cin_pad
    s.kh, s.kw, s.dil, s.stride = kh, kw, dil, stride
    s.pad = (dil * (kh - 1)) // 2 if pad is None else pad
    s.act = _lib.ACT[act]
    return s


def null_conv():
    return _lib.ConvW()
