"""Shared helpers of the parity tests: the product module and the oracle state_dict are filled
with the same hashed weights (oracle.cases / workloads.hashfill), nothing is read from disk."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import cases, refimport  # noqa: E402
from workloads import hashfill  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def make_cfg(C, impute=True, solver="euler", variable=True):
    return refimport.make_cfg(C, impute=impute, solver=solver, variable=variable)


def build_pair(C, solver="euler", impute=True, variable=True, delta_t=0.05, device="cuda"):
    """(product FuturePredictionODE on `device`, oracle state_dict on CPU) with identical weights."""
    import streamingflow_amd as sfa
    net = sfa.FuturePredictionODE(in_channels=C, latent_dim=C, n_future=4, cfg=make_cfg(C, impute, solver, variable),
                                  mixture=True, n_gru_blocks=2, n_res_layers=1, delta_t=delta_t).eval()
    sd = cases.fpode_state_dict(net.state_dict())
    net.load_state_dict(sd)
    if device != "cpu":
        net = net.to(device)
    return net, sd


def gold(name):
    return np.load(os.path.join(GOLD, name))


def maxabs(a, b):
    a = a.detach().cpu() if isinstance(a, torch.Tensor) else torch.from_numpy(np.asarray(a))
    b = b.detach().cpu() if isinstance(b, torch.Tensor) else torch.from_numpy(np.asarray(b))
    assert a.shape == b.shape, (a.shape, b.shape)
    return float((a.double() - b.double()).abs().max())
