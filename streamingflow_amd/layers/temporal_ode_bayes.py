"""MI355X-native GRU-ODE with Bayesian jumps (streamingflow/layers/temporal_ode_bayes.py).

``DualGRUODECell`` (:64-161), ``DualGRUCell`` (:211-305), ``GRUObservationCell`` (:308-344) and
``NNFOwithBayesianJumps`` (:355-627) with the reference constructor/forward signatures and
``state_dict`` keys; the arithmetic runs on libsfnative (HIP, gfx950).

Differences a caller can observe (all documented in DESIGN.md):
  * the step schedule is computed on the host up front (streamingflow_amd.schedule) — no
    device->host sync inside the rollout, which is enqueued as one C call / one hipGraph;
  * the Gaussian noise of ``infer_state`` is drawn for the whole rollout in one call
    (``torch.randn`` on the device) or supplied through ``self.noise`` (parity tests);
  * ``solver='rk4'`` is accepted in addition to the reference's 'euler' / 'midpoint';
  * inference only (no autograd), batch 1 per call as in the reference.
"""
import ctypes

import numpy as np
import torch
import torch.nn as nn

from .. import _lib, packing, runtime, schedule as sched
from ..runtime import PackedModule, ptr
from .convolutions import Bottleblock
from .res_models import ConvNet, SmallDecoder, SmallEncoder
from .temporal import pack_gru


class _DualCell(PackedModule):
    """Two conv-GRU branches mixed by a soft 'trusting gate'; 6 fused launches per evaluation."""
    derivative = False

    def __init__(self, input_size, hidden_size, gru_bias_init=0.0, norm='bn', activation='relu', bias=True):
        super().__init__()
        if input_size != hidden_size or hidden_size % 8 or hidden_size > 64:
            raise NotImplementedError("dual GRU cell: input_size == hidden_size, multiple of 8, <= 64")
        self.input_size, self.hidden_size, self.gru_bias_init = input_size, hidden_size, gru_bias_init
        c2 = input_size + hidden_size
        for tag in ("1", "2"):
            for name in ("conv_update_", "conv_reset_", "conv_state_tilde_"):
                setattr(self, name + tag, nn.Conv2d(c2 if tag == "1" else 2 * hidden_size, hidden_size, 3, padding=1))
        self.conv_decoder_2 = nn.Conv2d(hidden_size, hidden_size, kernel_size=3, bias=True, padding=1)
        self.trusting_gate = nn.Sequential(Bottleblock(2 * hidden_size, hidden_size),
                                           nn.Conv2d(hidden_size, 2, kernel_size=1, bias=False))

    def _pack(self):
        if self.gru_bias_init != 0.0:
            raise NotImplementedError("gru_bias_init != 0")
        C = self.hidden_size
        pk = packing.Pack(_lib.DualW())
        s = pk.struct
        g1 = pack_gru(pk, self.conv_update_1, self.conv_reset_1, self.conv_state_tilde_1, C, C)
        # gru_cell_2 is called as gru_cell_2(s, s): its gates see cat[s, s] -> duplicate input folded
        g2 = pack_gru(pk, self.conv_update_2, self.conv_reset_2, self.conv_state_tilde_2, C, C, fold_dup=True)
        s.gates1, s.cand1, s.gates2, s.cand2 = g1.gates, g1.cand, g2.gates, g2.cand
        s.dec2 = packing.conv_w(pk, self.conv_decoder_2.weight, C, bias=self.conv_decoder_2.bias)
        bb = self.trusting_gate[0]
        L = bb.layers
        s.tg7 = packing.conv_w(pk, L[0].weight, C, C, scale=L[1].weight, bias=L[1].bias)
        s.tg1 = packing.conv_w(pk, L[3].weight, C, scale=L[4].weight, bias=L[4].bias)
        s.tg3 = packing.conv_w(pk, L[6].weight, C, scale=L[7].weight, bias=L[7].bias)
        s.tgproj = packing.conv_w(pk, bb.projection[0].weight, C, C)
        s.w_logit = pk.hold(self.trusting_gate[1].weight.reshape(2, C))
        s.C = C
        return pk

    def run_nhwc(self, x, s, out, derivative, base=None, coef=None):
        h, w, C = s.shape[-3:]
        L = _lib.lib()
        ws = runtime.workspace(L.sf_dual_cell_ws_bytes(C, h, w), s.device)
        _lib.check(L.sf_dual_cell_fwd(self.packed().struct, ptr(x), ptr(s), ptr(out), int(derivative), ptr(base),
                                      ptr(coef), None, 0, h, w, ptr(ws), ws.numel() * 4,
                                      runtime.stream_ptr(s.device)), "dual_cell")
        return out

    def forward(self, x, state):
        runtime.require_cuda(x, state)
        squeeze5 = x.dim() == 5
        if squeeze5:
            if x.shape[1] != 1 or state.shape[1] != 1:
                raise NotImplementedError("n_present > 1 warm-up is unused by the reference forward path")
            x, state = x[:, 0], state[:, 0]
        if x.shape[0] != 1:
            raise NotImplementedError("batch 1 only (the reference mis-broadcasts for B > 1, SURVEY.md §0)")
        assert x.shape[1] == self.input_size, f'feature sizes must match, got input {x.shape[1]} for layer with size {self.input_size}'
        xn, sn = runtime.to_nhwc(x), runtime.to_nhwc(state)
        out = torch.empty_like(sn)
        if self.derivative:   # cur - s  ==  0 + 1*(cur - s)
            base = torch.zeros_like(sn)
            one = torch.ones(1, dtype=torch.float32, device=sn.device)
            self.run_nhwc(xn, sn, out, True, base, one)
        else:
            self.run_nhwc(xn, sn, out, False)
        return runtime.to_nchw(out)


class DualGRUODECell(_DualCell):
    """ODE derivative f(x, s) = cur - s (temporal_ode_bayes.py:92-131)."""
    derivative = True


class DualGRUCell(_DualCell):
    """Observation update, returns cur (temporal_ode_bayes.py:239-275)."""
    derivative = False


class GRUObservationCell(nn.Module):
    """temporal_ode_bayes.py:308-344: discrete update at an observation; `p` is ignored, loss None."""

    def __init__(self, input_size, hidden_size, min_log_sigma=-5.0, max_log_sigma=5.0, bias=True):
        super().__init__()
        self.gru_d = DualGRUCell(input_size, hidden_size, bias=bias)
        self.input_size, self.prep_hidden, self.var_eps = input_size, hidden_size, 1e-6
        self.min_log_sigma, self.max_log_sigma = min_log_sigma, max_log_sigma

    def forward(self, state, p, X_obs):
        return self.gru_d(X_obs, state), None


def init_weights(m):
    if type(m) == torch.nn.Linear:
        torch.nn.init.xavier_uniform_(m.weight)
        if m.bias is not None:
            m.bias.data.fill_(0.05)


class NNFOwithBayesianJumps(nn.Module):
    """Neural negative-feedback ODE with Bayesian jumps over BEV latents."""

    def __init__(self, input_size, hidden_size, cfg, bias=True, logvar=True, mixing=1, solver="euler",
                 min_log_sigma=-5.0, max_log_sigma=5.0, impute=False):
        super().__init__()
        self.impute = cfg.MODEL.IMPUTE          # ctor `impute` / `solver` are ignored, as in the reference (:365,:384)
        self.cfg = cfg
        self.min_log_sigma, self.max_log_sigma = min_log_sigma, max_log_sigma
        self.p_model = ConvNet(hidden_size, hidden_size * 2)
        self.gru_c = DualGRUODECell(input_size, hidden_size, bias=bias)
        self.gru_obs = GRUObservationCell(input_size, hidden_size, min_log_sigma=min_log_sigma,
                                          max_log_sigma=max_log_sigma, bias=bias)
        self.skipco = cfg.MODEL.SMALL_ENCODER.SKIPCO
        oc, fs = cfg.MODEL.ENCODER.OUT_CHANNELS, cfg.MODEL.SMALL_ENCODER.FILTER_SIZE
        self.srvp_encoder = SmallEncoder(oc, oc, fs)
        self.srvp_decoder = SmallDecoder(oc, oc, fs, self.skipco)
        self.solver = cfg.MODEL.SOLVER
        self.use_variable_ode_step = cfg.MODEL.FUTURE_PRED.USE_VARIABLE_ODE_STEP
        assert self.solver in ["euler", "midpoint", "rk4"], "Solver must be 'euler', 'midpoint' (reference) or 'rk4' (build-defined)."
        self.input_size, self.hidden_size, self.logvar, self.mixing = input_size, hidden_size, logvar, mixing
        self.noise = None    # None: torch.randn on the device; else callable(shape, dtype, device) -> NCHW eps per draw
        self.apply(init_weights)

    # ---- noise --------------------------------------------------------------------------------
    def _draw_eps(self, n_draws, h, w, device):
        """[n_draws, h, w, C] fp32 on `device`, one row per infer_state call in reference order."""
        C = self.hidden_size
        if self.noise is None:
            return torch.randn((max(1, n_draws), h, w, C), dtype=torch.float32, device=device)
        rows = [self.noise((1, C, h, w), torch.float32, "cpu") for _ in range(n_draws)]
        if not rows:
            return torch.zeros((1, h, w, C), dtype=torch.float32, device=device)
        return torch.cat(rows, 0).permute(0, 2, 3, 1).contiguous().to(device)

    # ---- reference API on NCHW tensors ------------------------------------------------------------
    def srvp_encode(self, x):
        b, t, c, h, w = x.shape
        hx = self.srvp_encoder(x.reshape(b * t, c, h, w))
        return hx.view(b, t, *hx.shape[1:]), None

    def srvp_decode(self, x, skip=None):
        b, t, c, h, w = x.shape
        out = self.srvp_decoder(x.reshape(b * t, c, h, w), skip=skip)
        return out.view(b, t, *out.shape[1:])

    def infer_state(self, x, deterministic=False):
        """(:463-477) returns (sample, raw p_model output).  `deterministic` is ignored as in the reference."""
        runtime.require_cuda(x)
        if x.shape[0] != 1:
            raise NotImplementedError("batch 1 only")
        sn = runtime.to_nhwc(x)
        _, h, w, C = sn.shape
        eps = self._draw_eps(1, h, w, x.device)
        p = torch.empty_like(sn)
        q = torch.empty((1, h, w, 2 * C), dtype=torch.float32, device=x.device)
        L = _lib.lib()
        ws = runtime.workspace(L.sf_infer_state_ws_bytes(C, h, w), x.device)
        _lib.check(L.sf_infer_state_fwd(self.p_model.packed().struct, ptr(sn), ptr(eps), ptr(p), ptr(q), h, w, ptr(ws),
                                        ws.numel() * 4, runtime.stream_ptr(x.device)), "infer_state")
        return runtime.to_nchw(p), runtime.to_nchw(q)

    def ode_step(self, state, input, delta_t, current_time):
        """(:436-461) one Euler / midpoint / RK4 step; returns the reference's 5-tuple."""
        runtime.require_cuda(state, input)
        if state.shape[0] != 1:
            raise NotImplementedError("batch 1 only")
        dev = state.device
        sn, pn = runtime.to_nhwc(state), runtime.to_nhwc(input)
        _, h, w, C = sn.shape
        sc = sched.Schedule(dts=[float(delta_t)])
        coef = torch.from_numpy(sc.coef_array()).to(dev)
        eps = self._draw_eps(sched.DRAWS_PER_STEP[self.solver], h, w, dev)
        s_out, p_out = torch.empty_like(sn), torch.empty_like(pn)
        L = _lib.lib()
        ws = runtime.workspace(L.sf_ode_step_ws_bytes(C, h, w), dev)
        _lib.check(L.sf_ode_step_fwd(self.gru_c.packed().struct, self.p_model.packed().struct,
                                     _lib.SOLVER[self.solver], int(bool(self.impute)), ptr(sn), ptr(pn), ptr(coef),
                                     ptr(eps), ptr(s_out), ptr(p_out), h, w, ptr(ws), ws.numel() * 4,
                                     runtime.stream_ptr(dev)), "ode_step")
        current_time = current_time + delta_t
        return (runtime.to_nchw(s_out), runtime.to_nchw(p_out), current_time,
                torch.tensor([0], device=dev, dtype=torch.float64), torch.tensor([0], device=dev, dtype=torch.float32))

    # ---- the rollout ----------------------------------------------------------------------------
    def rollout_nhwc(self, hx_obs, sc, eps=None):
        """hx_obs [n_obs, h, w, C] (encoded observations in time order), sc: Schedule.
        Returns (selected states [n_T, h, w, C], final state [h, w, C])."""
        n_obs, h, w, C = hx_obs.shape
        dev = hx_obs.device
        if eps is None:
            eps = self._draw_eps(sc.n_draws, h, w, dev)
        ops = sc.ops_array()
        sel = np.asarray(sc.sel_nops, dtype=np.int32)
        coef = torch.from_numpy(sc.coef_array()).to(dev)
        out = torch.empty((len(sel), h, w, C), dtype=torch.float32, device=dev)
        final = torch.empty((h, w, C), dtype=torch.float32, device=dev)
        L = _lib.lib()
        ws = runtime.workspace(L.sf_nnfo_rollout_ws_bytes(C, h, w), dev)
        _lib.check(L.sf_nnfo_rollout_fwd(
            self.gru_c.packed().struct, self.gru_obs.gru_d.packed().struct, self.p_model.packed().struct,
            _lib.SOLVER[self.solver], int(bool(self.impute)), ops.ctypes.data_as(_lib.i32p), len(sc.ops),
            ptr(hx_obs), ptr(eps), ptr(coef), sel.ctypes.data_as(_lib.i32p), len(sel), ptr(out), ptr(final), h, w,
            ptr(ws), ws.numel() * 4, runtime.stream_ptr(dev)), "nnfo_rollout")
        return out, final

    def forward_nhwc(self, times, obs_nhwc, delta_t, T):
        """obs_nhwc: [n_obs, H, W, C] observations sorted by `times`.  Returns (final latent state
        [h, w, C], decoded predictions [n_T, H', W', C])."""
        sc = sched.build_schedule([float(t) for t in times], [float(t) for t in T], delta_t,
                                  self.use_variable_ode_step, self.solver)
        hx = self.srvp_encoder.forward_nhwc(obs_nhwc)
        states, final = self.rollout_nhwc(hx, sc)
        return final, self.srvp_decoder.forward_nhwc(states), sc

    def forward(self, times, input, obs, delta_t, T, return_path=True):
        """(:479-627) times: 1-D float64 observation times (sorted), input: (1,1,C,H,W) (only its shape
        matters: the reference's encoding of it is overwritten before use), obs:
        (1,n_obs,C,H,W), T: 1-D float64 target times.  Returns (state, 0, x) as the reference does."""
        runtime.require_cuda(obs)
        if obs.shape[0] != 1:
            raise NotImplementedError("batch 1 only (as the reference)")
        final, x, _ = self.forward_nhwc(times, runtime.to_nhwc(obs[0]), delta_t, T)
        return runtime.to_nchw(final[None]), 0, runtime.to_nchw(x)[None]
