"""Portable closed-form tensor generator (shared by the benches and the checker; computes nothing of the path).

Weights, inputs and Gaussian noise for every parity test are regenerated on both sides (reference
import here, HIP path on the GPU box) from ``(name, shape, seed)`` with the same integer hash, so
no weights ever need to be committed (SURVEY.md §8c "Golden vectors").

splitmix64 over ``fnv1a64(name) ^ seed`` -> uniform [0,1) doubles -> scaled.  Pure numpy.
"""
import math

import numpy as np
import torch

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _fnv1a64(s: str) -> int:
    h = 0xCBF29CE484222325
    for ch in s.encode():
        h = ((h ^ ch) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def uniform01(name: str, n: int, seed: int = 0) -> np.ndarray:
    """n doubles in [0,1), a pure function of (name, seed, index)."""
    base = (_fnv1a64(name) ^ ((seed * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF)) & 0xFFFFFFFFFFFFFFFF
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) * np.uint64(0xD1342543DE82EF95) + np.uint64(base)
    bits = _splitmix64(idx) >> np.uint64(11)
    return bits.astype(np.float64) * (1.0 / 9007199254740992.0)


def uniform(name, shape, lo, hi, seed=0) -> torch.Tensor:
    n = int(np.prod(shape)) if len(shape) else 1
    u = uniform01(name, n, seed)
    return torch.from_numpy((lo + (hi - lo) * u).astype(np.float32)).reshape(tuple(shape))


def normal(name, shape, seed=0, std=1.0) -> torch.Tensor:
    """Box-Muller N(0, std^2) from two hashed uniforms."""
    n = int(np.prod(shape)) if len(shape) else 1
    u1 = uniform01(name + "#a", n, seed)
    u2 = uniform01(name + "#b", n, seed)
    z = np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(2.0 * math.pi * u2)
    return torch.from_numpy((std * z).astype(np.float32)).reshape(tuple(shape))


def fill_state_dict(sd, seed=0, gain=1.0):
    """Return a new dict with every tensor of ``sd`` replaced by a hashed one of the same shape.

    * ``running_mean`` U(-.2,.2), ``running_var`` U(.5,1.5), ``num_batches_tracked`` = 1
      (so that eval-mode BatchNorm folding is really exercised);
    * 1-D ``weight`` (BN / LayerNorm scale) U(.7,1.3); 1-D ``bias`` U(-.1,.1); ``gamma`` U(.3,.9);
    * >=2-D ``weight``: U(-a,a), a = gain*sqrt(3/fan_in), fan_in = numel/shape[0].
    """
    out = {}
    for k, v in sd.items():
        shape = tuple(v.shape)
        if k.endswith("num_batches_tracked"):
            out[k] = torch.ones_like(v)
        elif k.endswith("running_mean"):
            out[k] = uniform(k, shape, -0.2, 0.2, seed)
        elif k.endswith("running_var"):
            out[k] = uniform(k, shape, 0.5, 1.5, seed)
        elif k.endswith("gamma"):
            out[k] = uniform(k, shape, 0.3, 0.9, seed)
        elif v.dim() <= 1 and k.endswith("weight"):
            out[k] = uniform(k, shape, 0.7, 1.3, seed)
        elif v.dim() <= 1:
            out[k] = uniform(k, shape, -0.1, 0.1, seed)
        else:
            fan_in = v.numel() // shape[0]
            a = gain * math.sqrt(3.0 / fan_in)
            out[k] = uniform(k, shape, -a, a, seed)
        out[k] = out[k].to(v.dtype) if v.is_floating_point() else out[k].to(v.dtype)
    return out


class HashedNoise:
    """eps source: the k-th draw of shape S is ``normal(f'eps{k}', S, seed)`` (or zeros)."""

    def __init__(self, seed=0, zero=False):
        self.seed, self.zero, self.k = seed, zero, 0

    def __call__(self, shape, dtype=torch.float32, device="cpu"):
        k = self.k
        self.k += 1
        if self.zero:
            return torch.zeros(tuple(shape), dtype=dtype, device=device)
        return normal(f"eps{k}", tuple(shape), self.seed).to(dtype=dtype, device=device)
