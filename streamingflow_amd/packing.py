"""Load-time weight packing: reference-format parameters (OIHW fp32, BatchNorm buffers, ...) ->
the packed device layout of include/sfnative.h.  Runs once per (module, parameter version); uses
torch tensor ops on the parameters' own device as plumbing — nothing here is on the hot path.
"""
import ctypes

import torch

from . import _lib

# Math mode of the convolutions, decided when a module packs its weights:
#   "fp32"    exact fp32 on v_mfma_f32_16x16x4_f32 (default; every parity figure and the bench headline)
#   "bf16x3"  opt-in: operands split into two bf16 pieces, three v_mfma_f32_16x16x32_bf16 products, fp32 accumulators
#             (~1e-5 from the exact path; profiles/r03_bf16x3_*).  set_math_mode() re-packs modules on their next call.
_MATH_MODE = ["fp32"]


def set_math_mode(mode):
    if mode not in ("fp32", "bf16x3"):
        raise ValueError("math mode must be 'fp32' or 'bf16x3'")
    _MATH_MODE[0] = mode


def math_mode():
    return _MATH_MODE[0]


# Winograd F(2x2, 3x3) for the 3x3 / stride-1 layers of large launches (csrc/conv_wino.hip): exact fp32 arithmetic, 2.25x fewer
# multiplies, different rounding (<= 7e-6 on the BEV outputs of every BASELINE config: profiles/r04_winograd_accuracy_study.json).
# Decided when a module packs its weights (the transformed weights are a second copy); SF_WINO=0 or set_winograd(False) keeps
# every layer in the direct form.
import os as _os
_WINOGRAD = [_os.environ.get("SF_WINO", "1") != "0"]


def set_winograd(on):
    _WINOGRAD[0] = bool(on)


def winograd():
    return _WINOGRAD[0]


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
    """eval-mode BatchNorm after a conv -> per-channel (scale, bias) on the accumulator, computed by the library
    (``sf_bn_fold``: the device code sf_pack_conv itself folds with — the only implementation of the fold)."""
    from . import runtime
    f = lambda t: None if t is None else t.detach().to(torch.float32).contiguous()
    w, b, m, v, cb = f(bn.weight), f(bn.bias), f(bn.running_mean), f(bn.running_var), f(conv_bias)
    if not w.is_cuda:
        raise RuntimeError("weights are packed on the MI355X: move the module to the GPU first (no CPU fallback)")
    n = w.numel()
    sc, bi = torch.empty(n, dtype=torch.float32, device=w.device), torch.empty(n, dtype=torch.float32, device=w.device)
    _lib.check(_lib.lib().sf_bn_fold(runtime.ptr(cb), runtime.ptr(w), runtime.ptr(b), runtime.ptr(m), runtime.ptr(v), float(bn.eps), n,
                                     runtime.ptr(sc), runtime.ptr(bi), runtime.stream_ptr(w.device)), "bn_fold")
    for t in (w, b, m, v, cb):
        if t is not None:
            t.record_stream(torch.cuda.current_stream(w.device))      # a temporary (dtype cast) must outlive the launch
    return sc, bi


def conv_w(holder, weight, c0, c1=0, scale=None, bias=None, act="none", dil=1, stride=1, pad=None,
           transposed=False, fold_dup=False, interleave=False):
    """Pack one convolution with ``sf_pack_conv`` (csrc/pack.hip: layout change, ConvTranspose flip, duplicate-input
    fold, row interleave, zero padding — all on the device).  weight: Conv2d [cout][cin][kh][kw] (or ConvTranspose2d
    [cin][cout][kh][kw] with transposed=True, k3/s1/p1 only).  Returns a filled _lib.ConvW whose blob `holder` owns."""
    w = weight.detach().to(torch.float32).contiguous()
    if not w.is_cuda:
        raise RuntimeError("weights are packed on the MI355X: move the module to the GPU first (no CPU fallback)")
    if transposed:
        cin, cout, kh, kw = w.shape
    else:
        cout, cin, kh, kw = w.shape
        if fold_dup:
            cin //= 2
    assert cin == c0 + c1, (cin, c0, c1)
    flags = (_lib.PACK_TRANSPOSED if transposed else 0) | (_lib.PACK_FOLD_DUP if fold_dup else 0) | (_lib.PACK_INTERLEAVE if interleave else 0)
    if _MATH_MODE[0] == "bf16x3":
        flags |= _lib.PACK_BF16X3
    elif _WINOGRAD[0]:
        flags |= _lib.PACK_WINOGRAD      # the library adds the transformed copy where the layer qualifies (3x3, stride 1, ...)
    L = _lib.lib()
    nbytes = L.sf_pack_conv_bytes(cout, cin, kh, kw, flags)
    if nbytes == 0:
        raise ValueError(f"cannot pack a {tuple(w.shape)} convolution")
    blob = torch.empty(nbytes // 4, dtype=torch.float32, device=w.device)
    vec = lambda v: None if v is None else v.detach().to(device=w.device, dtype=torch.float32).contiguous()
    sc, bi = vec(scale), vec(bias)
    s = _lib.ConvW()
    from . import runtime
    _lib.check(L.sf_pack_conv(runtime.ptr(w), runtime.ptr(bi), runtime.ptr(sc), None, None, None, None, 0.0, cout, cin, kh, kw, c0, c1,
                              _lib.ACT[act], dil, stride, -1 if pad is None else pad, flags, runtime.ptr(blob), nbytes,
                              ctypes.byref(s), runtime.stream_ptr(w.device)), "pack_conv")
    holder.keep.append(blob)
    # the pack kernels read w / sc / bi asynchronously: a temporary among them (torch.cat of heads, a slice, a dtype cast) is
    # handed to the caching allocator only after the launches are ordered on this stream — it is NOT kept for the life of
    # the pack (a second resident copy of every such weight)
    for t in (w, sc, bi):
        if t is not None:
            t.record_stream(torch.cuda.current_stream(w.device))
    return s


def null_conv():
    return _lib.ConvW()


_FLOW = [None]
_FLOW_CHECK = [True]
FLOW_FALLBACKS = [0]      # rollouts of this process whose persistent launch timed out and that were re-run on the launch-per-layer path


def set_persistent_flow(on, check=True):
    """Opt in to (True) / out of (False) the persistent flow kernel for single-latent rollouts, or back to the SF_PERSIST
    environment default (None).  It needs an otherwise idle device: include/sfnative.h, sf_set_flow_mode.  Graphs captured
    before the switch keep the form they were captured in (NNFOwithBayesianJumps.drop_graphs() re-captures).

    ``check`` (default on): after every persistent rollout the host reads the kernel's count of timed-out dependency waits
    (``sf_flow_errors``: one stream synchronisation per rollout) and, when it is not zero, runs the same rollout again on the
    launch-per-layer path IN THIS PROCESS and counts it in ``FLOW_FALLBACKS`` — a device that turns out to be shared costs time,
    never a result.  ``check=False`` keeps the rollout asynchronous; a timeout then shows as NaN outputs (never as numbers)."""
    _FLOW[0] = None if on is None else bool(on)
    _FLOW_CHECK[0] = bool(check)
    return bool(_lib.lib().sf_set_flow_mode(-1 if on is None else int(bool(on))))


def flow_active():
    """Whether single-latent rollouts of this process run as the persistent flow kernel."""
    if _FLOW[0] is not None:
        return _FLOW[0]
    return _os.environ.get("SF_PERSIST", "0") not in ("", "0")


def flow_checked():
    return _FLOW_CHECK[0]
