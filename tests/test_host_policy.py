"""CPU: host-side policies that need no GPU."""
import pytest
import torch


def test_require_no_grad():
    from streamingflow_amd import runtime
    x = torch.zeros(2, 3)
    runtime.require_no_grad(x, None)
    xg = torch.zeros(2, 3, requires_grad=True)
    with pytest.raises(RuntimeError, match="inference-only"):
        runtime.require_no_grad(x, xg)
    with torch.no_grad():
        runtime.require_no_grad(xg)


def test_rollout_defaults_are_auto():
    import streamingflow_amd as sfa
    from util import make_cfg
    net = sfa.FuturePredictionODE(8, 8, 4, make_cfg(8))
    assert net.gru_ode.use_graph is None and net.gru_ode.in_kernel_noise is None and net.gru_ode.noise is None
