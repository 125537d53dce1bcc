"""Device plumbing shared by the modules: current HIP stream handle, a grow-only workspace per
device, NCHW<->NHWC conversion through the library, cached packed weights."""
import ctypes

import torch

from . import _lib

_WORKSPACES = {}


def stream_ptr(device=None):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("streamingflow_amd runs on MI355X only: got a %s tensor (no CPU fallback)" % t.device)


def workspace(nbytes, device):
    """Grow-only fp32 scratch buffer per (device, stream) (pointer is stable while it does not grow)."""
    dev = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    key = (dev, torch.cuda.current_stream(dev).cuda_stream)     # one scratch per stream: forwards on different streams may overlap
    cur = _WORKSPACES.get(key)
    need = (int(nbytes) + 3) // 4 + 1024
    if cur is None or cur.numel() < need:
        _WORKSPACES[key] = cur = torch.empty(need, dtype=torch.float32, device=torch.device("cuda", dev))
    return cur


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def require_no_grad(*tensors):
    """Inference only (the reference evaluates under torch.no_grad(), evaluate.py:117): the HIP path records no autograd
    history, so an input that asks for gradients while grad mode is on would silently get none — refuse it instead.
    Parameters with requires_grad=True are fine: calling a module outside no_grad() works and returns detached tensors
    ("Observable differences")."""
    if torch.is_grad_enabled():
        for t in tensors:
            if t is not None and t.requires_grad:
                raise RuntimeError("streamingflow_amd is inference-only: an input has requires_grad=True under enabled grad mode, "
                                   "but the HIP path records no autograd history (detach the input or use torch.no_grad())")


def f32c(t):
    return t.detach().to(torch.float32).contiguous()


def to_nhwc(x):
    """(n, C, H, W) fp32 cuda -> (n, H, W, C) contiguous (libsfnative transpose kernel)."""
    require_cuda(x)
    require_no_grad(x)      # every module entry point converts its NCHW inputs here
    x = f32c(x)
    n, c, h, w = x.shape
    out = torch.empty((n, h, w, c), dtype=torch.float32, device=x.device)
    if out.numel():
        _lib.check(_lib.lib().sf_nchw_to_nhwc(ptr(x), ptr(out), n, c, h * w, stream_ptr(x.device)), "nchw_to_nhwc")
    return out


def to_nchw(x):
    """(n, H, W, C) fp32 cuda -> (n, C, H, W) contiguous."""
    require_cuda(x)
    n, h, w, c = x.shape
    out = torch.empty((n, c, h, w), dtype=torch.float32, device=x.device)
    if out.numel():
        _lib.check(_lib.lib().sf_nhwc_to_nchw(ptr(x), ptr(out), n, c, h * w, stream_ptr(x.device)), "nhwc_to_nchw")
    return out


_PACK_GENERATION = [0]      # process-wide, monotonically increasing: never reused (unlike id() of a freed Pack)


def strip_runtime_state(d):
    """A module's ``__dict__`` without what only exists on this process's GPU: packed weight copies (ctypes structs full of device
    pointers), folded packs, captured graphs.  ``copy.deepcopy(model)`` and ``torch.save(model)`` go through ``__getstate__``; the
    copy re-packs (and re-captures) on its first forward."""
    d = dict(d)
    for k in [k for k in d if k.startswith("_sf_")] + ["_folded_tail"]:
        d.pop(k, None)
    return d


class PackedModule(torch.nn.Module):
    """Mixin: lazily packs the module's parameters for the HIP library and re-packs when any
    parameter was replaced, moved or modified in place (load_state_dict, .to(), optimiser).
    ``pack_generation()`` identifies the current packed copy for caches that hold raw pointers into it
    (captured hipGraphs): it changes exactly when ``packed()`` returns a new copy and is never reused."""

    def _param_signature(self):
        from . import packing
        sig = [packing.math_mode(), packing.winograd()]
        for t in list(self.parameters()) + list(self.buffers()):
            sig.append((t.data_ptr(), t._version, t.device))
        return tuple(sig)

    def packed(self):
        sig = self._param_signature()
        cache = self.__dict__.get("_sf_pack")
        if cache is None or cache[0] != sig:
            first = next(self.parameters())
            require_cuda(first)
            with torch.no_grad():
                pk = self._pack()
            _PACK_GENERATION[0] += 1
            self.__dict__["_sf_pack"] = cache = (sig, pk, _PACK_GENERATION[0])
        return cache[1]

    def pack_generation(self):
        self.packed()
        return self.__dict__["_sf_pack"][2]

    def _pack(self):
        raise NotImplementedError

    def __getstate__(self):
        return strip_runtime_state(self.__dict__)
