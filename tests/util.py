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


def oracle_rollout(sd, sc, hx_nhwc, eps_nhwc, solver="euler", impute=True, prefix="gru_ode"):
    """The oracle's GRU-ODE with Bayesian jumps at the LATENT level: a schedule (streamingflow_amd.schedule.Schedule — itself pinned to 44
    reference-captured schedules) applied with the oracle's own jump / step / infer_state restatements (oracle/ref_torch.py: dual_cell,
    ode_step, infer_state = temporal_ode_bayes.py:562-574 / :436-461 / :463-477) to encoded observations hx [n_obs, h, w, C]; eps
    [n_draws, h, w, C] is consumed in call order, as the reference's global generator would be.  Returns (selected states
    [n_T, h, w, C], final state [h, w, C]) in the product's NHWC layout.  Test infrastructure."""
    from oracle import ref_torch as R
    from streamingflow_amd._lib import OP_JUMP, OP_STEP
    hx = hx_nhwc.detach().cpu().permute(0, 3, 1, 2).contiguous()
    eps = eps_nhwc.detach().cpu().permute(0, 3, 1, 2).contiguous()
    draw = [0]

    def eps_fn(shape, dtype, device):
        e = eps[draw[0]][None]
        draw[0] += 1
        assert tuple(e.shape) == tuple(shape), (e.shape, shape)
        return e

    state = torch.zeros_like(hx[:1])
    inp = torch.zeros_like(hx[:1])
    after = [state]      # state after k ops
    with torch.no_grad():
        for kind, idx in sc.ops:
            if kind == OP_JUMP:
                state = R.dual_cell(sd, prefix + ".gru_obs.gru_d", hx[idx:idx + 1], state, False)
                inp = R.infer_state(sd, prefix, state, eps_fn)[0]
            else:
                assert kind == OP_STEP
                state, inp = R.ode_step(sd, prefix, state, inp, sc.dts[idx], solver, impute, eps_fn)
            after.append(state)
    sel = torch.cat([after[n] for n in sc.sel_nops], 0).permute(0, 2, 3, 1).contiguous()
    return sel, state[0].permute(1, 2, 0).contiguous()
