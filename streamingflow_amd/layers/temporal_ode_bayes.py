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
import collections
import ctypes

import numpy as np
import torch
import torch.nn as nn

from .. import _lib, packing, runtime, schedule as sched
from ..runtime import PackedModule, ptr
from .convolutions import Bottleblock
from .res_models import ConvNet, SmallDecoder, SmallEncoder
from .temporal import pack_gru


class _SingleGRU(PackedModule):
    """Single-branch conv-GRU cells of temporal_ode_bayes.py (``SpatialGRUODECell`` :14-61,
    ``SpatialGRUCell`` :165-208): gates = conv3x3 + bias + sigmoid, candidate = ConvBlock (conv3x3,
    BatchNorm, ReLU).  Defined by the reference but not instantiated on the shipped path."""
    ode = False

    def __init__(self, input_size, hidden_size, gru_bias_init=0.0, norm='bn', activation='relu', bias=True):
        super().__init__()
        if norm != 'bn' or activation != 'relu':
            raise NotImplementedError("conv + BatchNorm + ReLU candidate only")
        from ..beverse.basic_modules import ConvBlock
        self.input_size, self.hidden_size, self.bias, self.gru_bias_init = input_size, hidden_size, bias, gru_bias_init
        cat = input_size + hidden_size
        self.conv_update = nn.Conv2d(cat, hidden_size, kernel_size=3, bias=True, padding=1)
        self.conv_reset = nn.Conv2d(cat, hidden_size, kernel_size=3, bias=True, padding=1)
        self.conv_state_tilde = ConvBlock(cat, hidden_size, kernel_size=3, bias=False, norm=norm, activation=activation)

    def _pack(self):
        if self.training:
            raise RuntimeError("streamingflow_amd is inference-only: call .eval()")
        pk = packing.Pack(_lib.GruW())
        s = pk.struct
        wg = torch.cat([self.conv_update.weight, self.conv_reset.weight], 0)
        bg = torch.cat([self.conv_update.bias, self.conv_reset.bias], 0) + float(self.gru_bias_init)     # :54-55
        s.gates = packing.conv_w(pk, wg, self.input_size, self.hidden_size, bias=bg, act="sigmoid")
        sc, bi = packing.bn_fold(self.conv_state_tilde.norm)
        s.cand = packing.conv_w(pk, self.conv_state_tilde.conv.weight, self.input_size, self.hidden_size, scale=sc,
                                bias=bi, act="relu")
        return pk

    def forward(self, x, state):
        runtime.require_cuda(x, state)
        xn, sn = runtime.to_nhwc(x), runtime.to_nhwc(state)
        n, h, w, _ = xn.shape
        L = _lib.lib()
        ws = runtime.workspace(L.sf_gru_cell_ws_bytes(self.hidden_size, n, h, w), x.device)
        out = torch.empty_like(sn)
        fn = L.sf_gru_ode_cell_fwd if self.ode else L.sf_gru_cell_fwd
        _lib.check(fn(self.packed().struct, ptr(xn), ptr(sn), ptr(out), n, h, w, ptr(ws), ws.numel() * 4,
                      runtime.stream_ptr(x.device)), "single_gru_cell")
        return runtime.to_nchw(out)


class SpatialGRUODECell(_SingleGRU):
    """dh = u * (h~ - s) (temporal_ode_bayes.py:35-61)."""
    ode = True


class SpatialGRUCell(_SingleGRU):
    """(1 - u) * s + u * h~ (temporal_ode_bayes.py:184-208)."""
    ode = False


class _DualCell(PackedModule):
    """Two conv-GRU branches mixed by a soft 'trusting gate'; 6 fused launches per evaluation."""
    derivative = False

    def __init__(self, input_size, hidden_size, gru_bias_init=0.0, norm='bn', activation='relu', bias=True):
        super().__init__()
        if input_size != hidden_size or hidden_size % 8 or hidden_size > 128:
            raise NotImplementedError("dual GRU cell: input_size == hidden_size, multiple of 8, <= 128")
        self.input_size, self.hidden_size, self.gru_bias_init = input_size, hidden_size, gru_bias_init
        c2 = input_size + hidden_size
        for tag in ("1", "2"):
            for name in ("conv_update_", "conv_reset_", "conv_state_tilde_"):
                setattr(self, name + tag, nn.Conv2d(c2 if tag == "1" else 2 * hidden_size, hidden_size, 3, padding=1))
        self.conv_decoder_2 = nn.Conv2d(hidden_size, hidden_size, kernel_size=3, bias=True, padding=1)
        self.trusting_gate = nn.Sequential(Bottleblock(2 * hidden_size, hidden_size),
                                           nn.Conv2d(hidden_size, 2, kernel_size=1, bias=False))

    def _pack(self):
        C = self.hidden_size
        pk = packing.Pack(_lib.DualW())
        s = pk.struct
        gb = self.gru_bias_init       # added to the gate pre-activations of both cells (:139-140, :154-155)
        g1 = pack_gru(pk, self.conv_update_1, self.conv_reset_1, self.conv_state_tilde_1, C, C, gate_bias=gb)
        # gru_cell_2 is called as gru_cell_2(s, s): its gates see cat[s, s] -> duplicate input folded
        g2 = pack_gru(pk, self.conv_update_2, self.conv_reset_2, self.conv_state_tilde_2, C, C, fold_dup=True, gate_bias=gb)
        s.gates1, s.cand1, s.gates2, s.cand2 = g1.gates, g1.cand, g2.gates, g2.cand
        # the two input halves of gates1 on their own (rollout: the state half is accumulated beside the previous infer_state)
        wg = torch.cat([self.conv_update_1.weight, self.conv_reset_1.weight], 0)
        bg = torch.cat([self.conv_update_1.bias, self.conv_reset_1.bias], 0) + float(gb)
        s.gates1_x = packing.conv_w(pk, wg[:, :C], C, bias=bg, act="sigmoid")
        s.gates1_s = packing.conv_w(pk, wg[:, C:], C)
        s.dec2 = packing.conv_w(pk, self.conv_decoder_2.weight, C, bias=self.conv_decoder_2.bias)
        bb = self.trusting_gate[0]
        L = bb.layers
        s.tg7 = packing.conv_w(pk, L[0].weight, C, C, scale=L[1].weight, bias=L[1].bias)
        # its two input halves on their own (rollout: the rnn_state2 half is accumulated on a forked stream beside the previous infer_state)
        s.tg7_h = packing.conv_w(pk, L[0].weight[:, :C], C, scale=L[1].weight, bias=L[1].bias)
        s.tg7_r = packing.conv_w(pk, L[0].weight[:, C:], C)
        s.tg1 = packing.conv_w(pk, L[3].weight, C, scale=L[4].weight, bias=L[4].bias)
        s.tg3 = packing.conv_w(pk, L[6].weight, C, scale=L[7].weight, bias=L[7].bias)
        s.tgproj = packing.conv_w(pk, bb.projection[0].weight, C, C)
        s.w_logit = pk.hold(self.trusting_gate[1].weight.reshape(2, C))
        s.C = C
        return pk

    def run_nhwc(self, x, s, out, derivative, base=None, coef=None):
        """x, s, out: [B, h, w, C] (B samples processed as one pixel space)."""
        B, h, w, C = s.shape
        L = _lib.lib()
        ws = runtime.workspace(L.sf_dual_cell_ws_bytes(C, B, h, w), s.device)
        _lib.check(L.sf_dual_cell_fwd(self.packed().struct, ptr(x), ptr(s), ptr(out), int(derivative), ptr(base),
                                      ptr(coef), None, 0, B, h, w, ptr(ws), ws.numel() * 4,
                                      runtime.stream_ptr(s.device)), "dual_cell")
        return out

    def _general_pack(self):
        """Second packed copy for calls with several present frames: cell 2 then sees two different tensors, so its gate
        convolution keeps both input halves (the main copy folds them, `_pack`)."""
        sig = self._param_signature()
        cache = self.__dict__.get("_sf_general")
        if cache is None or cache[0] != sig:
            C, gb = self.hidden_size, self.gru_bias_init
            pk = packing.Pack(None)
            with torch.no_grad():
                g1 = pack_gru(pk, self.conv_update_1, self.conv_reset_1, self.conv_state_tilde_1, C, C, gate_bias=gb)
                g2 = pack_gru(pk, self.conv_update_2, self.conv_reset_2, self.conv_state_tilde_2, C, C, gate_bias=gb)
                dec2 = packing.conv_w(pk, self.conv_decoder_2.weight, C, bias=self.conv_decoder_2.bias)
            pk.struct = (g1, g2, dec2)
            self.__dict__["_sf_general"] = cache = (sig, pk)
        return cache[1].struct

    def _forward_frames(self, x, state):
        """state [b, n_present > 1, C, h, w] (temporal_ode_bayes.py:101-131 / :248-275): branch 2's hidden state starts from
        the first frame (the ODE cell warms it up over the present frames), both branches start from the last one.
        Unused by the reference's forward path."""
        from .temporal import conv_nhwc, gru_cell_nhwc, trust_mix_nhwc
        g1, g2, dec2 = self._general_pack()
        n = state.shape[1]
        frames = [runtime.to_nhwc(state[:, t]) for t in range(n)]
        hid = frames[0]
        if self.derivative:               # only the ODE cell warms branch 2 up (:106-109); the observation cell keeps state[:, 0] (:252)
            for t in range(n - 1):
                hid = gru_cell_nhwc(g2, frames[t], hid)
        r1 = gru_cell_nhwc(g1, runtime.to_nhwc(x[:, 0]), frames[-1])
        hid = gru_cell_nhwc(g2, frames[-1], hid)
        r2 = conv_nhwc(dec2, hid)
        cur = runtime.to_nchw(trust_mix_nhwc(self.packed().struct, r1, r2))
        if not self.derivative:
            return cur
        # the reference returns `cur_state - state.squeeze(1)`: with several frames the squeeze is a no-op and the
        # difference broadcasts over them (one sample)
        if state.shape[0] != 1:
            raise ValueError("DualGRUODECell with n_present > 1: one sample per call (the reference's broadcast)")
        return cur[:, None] - state

    def forward(self, x, state):
        runtime.require_cuda(x, state)
        if x.dim() == 5:
            assert x.shape[2] == self.input_size, f'feature sizes must match, got input {x.shape[2]} for layer with size {self.input_size}'
            if state.shape[1] != 1:
                return self._forward_frames(x, state)
            x, state = x[:, 0], state[:, 0]
        assert x.shape[1] == self.input_size, f'feature sizes must match, got input {x.shape[1]} for layer with size {self.input_size}'
        # note: the reference mis-broadcasts for batch > 1; here every sample of the
        # batch gets the batch-1 semantics
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
        if self.skipco:      # res_models.py:98-109, :134-147 — not on the shipped path; never silently ignored
            raise NotImplementedError("MODEL.SMALL_ENCODER.SKIPCO=True (encoder -> decoder skip connections) is not built")
        oc, fs = cfg.MODEL.ENCODER.OUT_CHANNELS, cfg.MODEL.SMALL_ENCODER.FILTER_SIZE
        self.srvp_encoder = SmallEncoder(oc, oc, fs)
        self.srvp_decoder = SmallDecoder(oc, oc, fs, self.skipco)
        self.solver = cfg.MODEL.SOLVER
        self.use_variable_ode_step = cfg.MODEL.FUTURE_PRED.USE_VARIABLE_ODE_STEP
        assert self.solver in ["euler", "midpoint", "rk4"], "Solver must be 'euler', 'midpoint' (reference) or 'rk4' (build-defined)."
        self.input_size, self.hidden_size, self.logvar, self.mixing = input_size, hidden_size, logvar, mixing
        self.noise = None    # None: torch.randn on the device; else callable(shape, dtype, device) -> NCHW eps per draw
        # throughput mode: the noise of infer_state is generated inside the sampling epilogue (Philox4x32-10 keyed by
        # (noise_seed, call counter)); no eps tensor exists.  Same distribution, its own stream; ignored when eps is given
        # None = auto: on whenever no `noise` source was injected (nothing to replay), True / False force it
        self.in_kernel_noise = None
        # None = derived at the first draw from torch's global seed (torch.manual_seed reaches the in-kernel noise as it reaches the
        # reference's torch.randn) mixed with the distributed rank and a per-module serial number: ranks and module instances draw
        # different streams.  `seed_noise(seed)` pins it (and restarts the call counter); neither is part of state_dict.
        self.noise_seed = None
        self._noise_calls = 0
        NNFOwithBayesianJumps._serial = getattr(NNFOwithBayesianJumps, "_serial", 0) + 1
        self._noise_serial = NNFOwithBayesianJumps._serial
        # capture the rollout of each schedule structure into a hipGraph and replay it.  None = auto: on for rollouts that
        # run on the launch-bound single-latent kernels (B*h*w < 4096), where ~9 short launches per step are replayed from one
        # graph; the results are cloned out of the graph's static buffers.  True: always, and the returned tensors ARE the
        # static buffers (valid until the next replay); False: never
        self.use_graph = None
        # at most GRAPH_CACHE_MAX captured rollouts are kept, least recently used first out (a stream of variable timestamps produces a
        # new schedule structure per call: each entry pins its own buffers and a rollout workspace); when auto mode sees more than
        # GRAPH_AUTO_MAX_STRUCTURES distinct structures it stops capturing and runs eagerly
        self._graphs = collections.OrderedDict()
        self._graph_structures_seen = set()
        self._graph_gens = None
        self.apply(init_weights)

    # ---- noise --------------------------------------------------------------------------------
    def _draw_eps(self, n_draws, B, h, w, device):
        """[n_draws, B, h, w, C] fp32 on `device`, one row per infer_state call in reference order."""
        C = self.hidden_size
        if self.noise is None:
            return torch.randn((max(1, n_draws), B, h, w, C), dtype=torch.float32, device=device)
        rows = [self.noise((B, C, h, w), torch.float32, "cpu") for _ in range(n_draws)]
        if not rows:
            return torch.zeros((1, B, h, w, C), dtype=torch.float32, device=device)
        return torch.stack(rows, 0).permute(0, 1, 3, 4, 2).contiguous().to(device)

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
        sn = runtime.to_nhwc(x)
        B, h, w, C = sn.shape
        eps = self._draw_eps(1, B, h, w, x.device)
        p = torch.empty_like(sn)
        q = torch.empty((B, h, w, 2 * C), dtype=torch.float32, device=x.device)
        L = _lib.lib()
        ws = runtime.workspace(L.sf_infer_state_ws_bytes(C, B, h, w), x.device)
        _lib.check(L.sf_infer_state_fwd(self.p_model.packed().struct, ptr(sn), ptr(eps), ptr(p), ptr(q), B, h, w,
                                        ptr(ws), ws.numel() * 4, runtime.stream_ptr(x.device)), "infer_state")
        return runtime.to_nchw(p), runtime.to_nchw(q)

    def ode_step(self, state, input, delta_t, current_time):
        """(:436-461) one Euler / midpoint / RK4 step; returns the reference's 5-tuple."""
        runtime.require_cuda(state, input)
        dev = state.device
        sn, pn = runtime.to_nhwc(state), runtime.to_nhwc(input)
        B, h, w, C = sn.shape
        sc = sched.Schedule(dts=[float(delta_t)])
        coef = torch.from_numpy(sc.coef_array()).to(dev)
        eps = self._draw_eps(sched.DRAWS_PER_STEP[self.solver], B, h, w, dev)
        s_out, p_out = torch.empty_like(sn), torch.empty_like(pn)
        L = _lib.lib()
        ws = runtime.workspace(L.sf_ode_step_ws_bytes(C, B, h, w), dev)
        _lib.check(L.sf_ode_step_fwd(self.gru_c.packed().struct, self.p_model.packed().struct,
                                     _lib.SOLVER[self.solver], int(bool(self.impute)), ptr(sn), ptr(pn), ptr(coef),
                                     ptr(eps), ptr(s_out), ptr(p_out), B, h, w, ptr(ws), ws.numel() * 4,
                                     runtime.stream_ptr(dev)), "ode_step")
        current_time = current_time + delta_t
        return (runtime.to_nchw(s_out), runtime.to_nchw(p_out), current_time,
                torch.tensor([0], device=dev, dtype=torch.float64), torch.tensor([0], device=dev, dtype=torch.float32))

    # ---- the rollout ----------------------------------------------------------------------------
    def _enqueue_rollout(self, s0, per_image, hx_obs, eps, coef, out, final, ws, B, h, w, philox=None):
        ops = s0.ops_array()
        sel = np.asarray(s0.sel_nops, dtype=np.int32)
        L = _lib.lib()
        if philox is not None:
            _lib.check(L.sf_nnfo_rollout_philox_fwd(
                self.gru_c.packed().struct, self.gru_obs.gru_d.packed().struct, self.p_model.packed().struct,
                _lib.SOLVER[self.solver], int(bool(self.impute)), ops.ctypes.data_as(_lib.i32p), len(s0.ops),
                ptr(hx_obs), ptr(philox), ptr(coef), int(per_image), sel.ctypes.data_as(_lib.i32p), len(sel), ptr(out),
                ptr(final), B, h, w, ptr(ws), ws.numel() * 4, runtime.stream_ptr(hx_obs.device)), "nnfo_rollout_philox")
            return
        _lib.check(L.sf_nnfo_rollout_fwd(
            self.gru_c.packed().struct, self.gru_obs.gru_d.packed().struct, self.p_model.packed().struct,
            _lib.SOLVER[self.solver], int(bool(self.impute)), ops.ctypes.data_as(_lib.i32p), len(s0.ops),
            ptr(hx_obs), ptr(eps), ptr(coef), int(per_image), sel.ctypes.data_as(_lib.i32p), len(sel), ptr(out),
            ptr(final), B, h, w, ptr(ws), ws.numel() * 4, runtime.stream_ptr(hx_obs.device)), "nnfo_rollout")

    GRAPH_CACHE_MAX = 4
    GRAPH_AUTO_MAX_STRUCTURES = 16

    def seed_noise(self, seed):
        """Pin the seed of the in-kernel (Philox) noise and restart its call counter."""
        self.noise_seed = int(seed) & 0x7FFFFFFFFFFFFFFF
        self._noise_calls = 0
        self._noise_pinned = True

    def _philox_seed(self):
        if self.noise_seed is None:
            rank = torch.distributed.get_rank() if torch.distributed.is_available() and torch.distributed.is_initialized() else 0
            x = (int(torch.initial_seed()) * 0x9E3779B97F4A7C15 + rank * 0xBF58476D1CE4E5B9 + self._noise_serial * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
            x ^= x >> 31
            self.noise_seed = x & 0x7FFFFFFFFFFFFFFF
        return self.noise_seed

    def __getstate__(self):
        # captured graphs (hipGraphExec handles + their static buffers) belong to this object alone: a copy starts without them
        d = runtime.strip_runtime_state(self.__dict__)
        d["_graphs"] = collections.OrderedDict()
        d["_graph_structures_seen"] = set()
        d["_graph_gens"] = None
        # a copy (copy.deepcopy, an EMA twin, torch.save / load) draws its OWN Philox stream: new instance serial, seed derived again on
        # first use, call counter restarted — unless seed_noise() pinned the seed, which a copy keeps on purpose
        if not d.get("_noise_pinned", False):
            NNFOwithBayesianJumps._serial += 1
            d["_noise_serial"] = NNFOwithBayesianJumps._serial
            d["noise_seed"] = None
            d["_noise_calls"] = 0
        return d

    def drop_graphs(self):
        """Destroy every captured rollout graph of this module and release its static buffers."""
        L = _lib.lib()
        for g in self._graphs.values():
            if g.get("exec") is not None:
                L.sf_graph_destroy(g["exec"])
        self._graphs.clear()
        self._graph_structures_seen.clear()

    def rollout_nhwc(self, hx_obs, sc, eps=None):
        """hx_obs: [n_obs, B, h, w, C] (or [n_obs, h, w, C] for one sample) encoded observations in
        time order; sc: one Schedule shared by all samples, or a list of B Schedules with the same
        structure (``Schedule.key()``) and per-sample step sizes.
        Returns (selected states [n_T, B, h, w, C], final state [B, h, w, C]); without the B axis
        when hx_obs had none.  With ``self.use_graph`` the whole rollout (every kernel of every
        step and jump) is captured once per (schedule structure, shape) into a hipGraph and
        replayed; step sizes, observations and noise are fed through static device buffers."""
        one = hx_obs.dim() == 4
        if one:
            hx_obs = hx_obs[:, None]
        n_obs, B, h, w, C = hx_obs.shape
        dev = hx_obs.device
        scs = list(sc) if isinstance(sc, (list, tuple)) else [sc]
        s0 = scs[0]
        if len(scs) not in (1, B) or any(x.key() != s0.key() for x in scs):
            raise ValueError("batched rollout needs one schedule, or one per sample with identical structure")
        per_image = len(scs) > 1 and any(x.dts != s0.dts for x in scs)
        # draws the kernels will read: one per jump, DRAWS_PER_STEP[solver] per step (a schedule built for another solver
        # would make them read past the end of eps)
        need = s0.n_jumps + sched.DRAWS_PER_STEP[self.solver] * s0.n_steps
        philox = None
        in_kernel = self.noise is None if self.in_kernel_noise is None else bool(self.in_kernel_noise)
        auto_graph = self.use_graph is None and B * h * w < 4096
        if auto_graph:
            # not from inside somebody else's capture (warm-up, synchronize and a nested capture would break it), and not for an
            # endless variety of schedule structures
            # (the set stops growing at the cap: a stream of ever new timestamp structures then runs eagerly, while structures whose graph is
            # still cached keep replaying it)
            if len(self._graph_structures_seen) <= self.GRAPH_AUTO_MAX_STRUCTURES:
                self._graph_structures_seen.add(s0.key())
            cached = any(k[0] == s0.key() for k in self._graphs)
            if torch.cuda.is_current_stream_capturing() or (len(self._graph_structures_seen) > self.GRAPH_AUTO_MAX_STRUCTURES and not cached):
                auto_graph = False
        if eps is None and in_kernel and self.noise is None:
            self._noise_calls += 1
            philox = torch.tensor([self._philox_seed(), self._noise_calls], dtype=torch.int64, device=dev)
            eps = torch.empty((0, B, h, w, C), dtype=torch.float32, device=dev)      # placeholder (shape key of the graph cache)
        elif eps is None:
            eps = self._draw_eps(need, B, h, w, dev)
        elif eps.dim() == 4:
            eps = eps[:, None]
        if philox is None and (eps.shape[0] < need or tuple(eps.shape[1:]) != (B, h, w, C)):
            raise ValueError(f"eps must be [{need}, {B}, {h}, {w}, {C}] for solver {self.solver!r}, got {tuple(eps.shape)}")
        coef_np = np.stack([x.coef_array() for x in scs], axis=1) if per_image else s0.coef_array()
        coef = torch.from_numpy(np.ascontiguousarray(coef_np)).to(dev)
        L = _lib.lib()
        nbytes = L.sf_nnfo_rollout_ws_bytes(C, B, h, w)
        if not (self.use_graph or auto_graph):
            out = torch.empty((len(s0.sel_nops), B, h, w, C), dtype=torch.float32, device=dev)
            final = torch.empty((B, h, w, C), dtype=torch.float32, device=dev)
            ws = runtime.workspace(nbytes, dev)
            self._enqueue_rollout(s0, per_image, hx_obs, eps, coef, out, final, ws, B, h, w, philox)
        else:
            # a captured graph holds raw pointers into the packed weights: when any of the three packs was rebuilt
            # (load_state_dict, .to(), in-place update) every cached graph is stale — destroy them and their buffers
            gens = (self.gru_c.pack_generation(), self.p_model.pack_generation(), self.gru_obs.gru_d.pack_generation())
            if self._graph_gens != gens:
                self.drop_graphs()
                self._graph_gens = gens
            from .. import packing
            key = (s0.key(), per_image, tuple(hx_obs.shape), tuple(eps.shape), str(dev), self.solver, bool(self.impute), philox is not None,
                   packing._FLOW[0])      # a graph keeps the form (launch per layer / persistent flow) it was captured in
            g = self._graphs.get(key)
            if g is None:
                g = {"hx": torch.empty_like(hx_obs), "eps": torch.empty_like(eps), "coef": torch.empty_like(coef),
                     "out": torch.empty((len(s0.sel_nops), B, h, w, C), dtype=torch.float32, device=dev),
                     "final": torch.empty((B, h, w, C), dtype=torch.float32, device=dev),
                     "ws": torch.empty(nbytes // 4 + 1024, dtype=torch.float32, device=dev),
                     "philox": torch.empty_like(philox) if philox is not None else None}
                g["hx"].copy_(hx_obs); g["eps"].copy_(eps); g["coef"].copy_(coef)
                if philox is not None:
                    g["philox"].copy_(philox)
                # eager warm-up (sets kernel attributes, packs weights), then capture on a side stream
                self._enqueue_rollout(s0, per_image, g["hx"], g["eps"], g["coef"], g["out"], g["final"], g["ws"], B, h, w, g["philox"])
                torch.cuda.synchronize(dev)
                side = torch.cuda.Stream(device=dev)
                with torch.cuda.stream(side):
                    sp = runtime.stream_ptr(dev)
                    _lib.check(L.sf_graph_begin(sp), "graph_begin")
                    try:
                        self._enqueue_rollout(s0, per_image, g["hx"], g["eps"], g["coef"], g["out"], g["final"], g["ws"], B, h, w, g["philox"])
                    finally:
                        ex = ctypes.c_void_p()
                        _lib.check(L.sf_graph_end(sp, ctypes.byref(ex)), "graph_end")
                g["exec"] = ex
                self._graphs[key] = g
                while len(self._graphs) > self.GRAPH_CACHE_MAX:      # least recently used out: its graph is destroyed, its buffers released
                    _, old_g = self._graphs.popitem(last=False)
                    if old_g.get("exec") is not None:
                        L.sf_graph_destroy(old_g["exec"])
            else:
                self._graphs.move_to_end(key)
            g["hx"].copy_(hx_obs); g["eps"].copy_(eps); g["coef"].copy_(coef)
            if philox is not None:
                g["philox"].copy_(philox)
            _lib.check(L.sf_graph_launch(g["exec"], runtime.stream_ptr(dev)), "graph_launch")
            out, final = g["out"], g["final"]      # valid until the next replay of this graph
            if auto_graph:                          # nobody asked for the zero-copy form: hand out fresh tensors
                out, final = out.clone(), final.clone()
        from .. import packing as _pk
        if _pk.flow_active() and _pk.flow_checked() and not torch.cuda.is_current_stream_capturing():
            # persistent flow kernel: a dependency wait that gave up (device shared with another stream / process) must cost time, not a
            # result — the same rollout again on the launch-per-layer path, in this process, counted
            # (read from THIS rollout's own output — the poison kernel turns every element into NaN — not from sf_flow_errors, whose
            # "most recent rollout of the thread" is the most recent one ENQUEUED, not the graph that was just replayed)
            if bool(torch.isnan(final.reshape(-1)[:1]).item()):
                _pk.FLOW_FALLBACKS[0] += 1
                was = L.sf_set_flow_mode(0)
                try:
                    if philox is None:
                        eps_c, hx_c = eps.contiguous(), hx_obs.contiguous()
                    else:
                        eps_c, hx_c = eps, hx_obs.contiguous()
                    out = torch.empty((len(s0.sel_nops), B, h, w, C), dtype=torch.float32, device=dev)
                    final = torch.empty((B, h, w, C), dtype=torch.float32, device=dev)
                    ws2 = runtime.workspace(nbytes, dev)
                    self._enqueue_rollout(s0, per_image, hx_c, eps_c, coef, out, final, ws2, B, h, w, philox)
                finally:
                    L.sf_set_flow_mode(int(bool(was)) if _pk._FLOW[0] is not None else -1)
        return (out[:, 0], final[0]) if one else (out, final)

    def make_schedule(self, times, delta_t, T):
        return sched.build_schedule([float(t) for t in times], [float(t) for t in T], delta_t,
                                    self.use_variable_ode_step, self.solver)

    def forward_nhwc(self, scs, obs_nhwc):
        """obs_nhwc: [n_obs, B, H, W, C] observations in time order; scs: Schedule or list of B
        Schedules.  Returns (final latent state [B, h, w, C], decoded predictions [n_T, B, H', W', C])."""
        n_obs, B, H, W, C = obs_nhwc.shape
        hx = self.srvp_encoder.forward_nhwc(obs_nhwc.reshape(n_obs * B, H, W, C))
        hx = hx.view(n_obs, B, *hx.shape[1:])
        states, final = self.rollout_nhwc(hx, scs)
        n_T = states.shape[0]
        x = self.srvp_decoder.forward_nhwc(states.reshape(n_T * B, *states.shape[2:]))
        return final, x.view(n_T, B, *x.shape[1:])

    def forward(self, times, input, obs, delta_t, T, return_path=True):
        """(:479-627) times: 1-D float64 observation times (sorted), input: (1,1,C,H,W) (only its shape
        matters: the reference's encoding of it is overwritten before use), obs:
        (1,n_obs,C,H,W), T: 1-D float64 target times.  Returns (state, 0, x) as the reference does."""
        runtime.require_cuda(obs)
        if obs.shape[0] != 1:
            raise NotImplementedError("one sample per call (as the reference); FuturePredictionODE batches samples")
        sc = self.make_schedule(times, delta_t, T)
        final, x = self.forward_nhwc(sc, runtime.to_nhwc(obs[0])[:, None])
        return runtime.to_nchw(final), 0, runtime.to_nchw(x[:, 0])[None]
