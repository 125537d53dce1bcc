"""TEST INFRASTRUCTURE — CPU restatement (plain torch fp32) of the reference's GRU-ODE hot path.

This is the oracle the HIP path is checked against (and the "port" CPU baseline in bench.py).
It is NOT product code: ``streamingflow_amd`` never imports it.  It is written functionally over
a reference-format ``state_dict`` (``sd``, a dict name -> tensor, reference key names) so that a
real StreamingFlow checkpoint slice (``model.future_prediction_ode.*``) can be fed to it directly.

Pinning: the reference has no tests for this path; this file is validated against the reference
itself, imported from /root/reference by ``oracle/gen_golden.py`` / ``tests/test_oracle_vs_reference.py``
(<=1e-6 max-abs per op) and against the committed fixtures in ``tests/golden``.

All ``file:line`` citations are relative to the reference root.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------------------------
# primitives
# --------------------------------------------------------------------------------------------
def _conv(sd, p, x, padding=0, dilation=1, stride=1, groups=1):
    return F.conv2d(x, sd[p + ".weight"], sd.get(p + ".bias"), stride, padding, dilation, groups)


def _bn(sd, p, x):
    # eval-mode BatchNorm2d, eps 1e-5 (torch default)
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"],
                        sd[p + ".bias"], False, 0.0, 1e-5)


def _act(x, kind):
    if kind == "lrelu":
        return F.leaky_relu(x, 0.1)
    if kind == "relu":
        return F.relu(x)
    if kind == "tanh":
        return torch.tanh(x)
    if kind == "none":
        return x
    raise ValueError(kind)


def rm_conv_block(sd, p, x, act="lrelu", norm=True, transpose=False):
    """layers/res_models.py:8-49 ``ConvBlock`` (k=3, stride 1, pad 1)."""
    if transpose:
        x = F.conv_transpose2d(x, sd[p + ".conv.weight"], sd.get(p + ".conv.bias"), 1, 1)
    else:
        x = _conv(sd, p + ".conv", x, padding=1)
    if norm:
        x = _bn(sd, p + ".norm", x)
    return _act(x, act)


def res_block(sd, p, x):
    """layers/res_models.py:52-79 ``ResBlock`` (Dropout2d inert in eval)."""
    r = rm_conv_block(sd, p + ".layers.conv_1", x)
    r = rm_conv_block(sd, p + ".layers.conv_2", r)
    if (p + ".projection.weight") in sd:
        x = _conv(sd, p + ".projection", x)
    return x + r


def small_encoder(sd, p, x):
    """layers/res_models.py:98-109 (skips unused, SKIPCO=False)."""
    h = x
    for i in range(5):
        if i in (1, 2):
            h = F.max_pool2d(h, 2, 2)
        h = res_block(sd, f"{p}.blocks.{i}", h)
    return rm_conv_block(sd, p + ".last_conv.0", h, act="tanh")


def small_decoder(sd, p, z):
    """layers/res_models.py:134-147 with skip=None."""
    h = rm_conv_block(sd, p + ".first_upconv", z, transpose=True)
    for i in range(5):
        h = res_block(sd, f"{p}.blocks.{i}", h)
        if i in (2, 3):
            h = F.interpolate(h, scale_factor=2, mode="nearest")
    h = rm_conv_block(sd, p + ".last_conv.0", h)
    return rm_conv_block(sd, p + ".last_conv.1", h, norm=False, transpose=True)


def se_layer(sd, p, x):
    """layers/res_models.py:161-165."""
    y = x.mean(dim=(2, 3))
    y = torch.sigmoid(F.linear(F.relu(F.linear(y, sd[p + ".fc.0.weight"])), sd[p + ".fc.2.weight"]))
    return x * y[:, :, None, None]


def conv_net(sd, p, x):
    """layers/res_models.py:168-180 ``ConvNet`` (= p_model)."""
    x = res_block(sd, p + ".model.0", x)
    x = se_layer(sd, p + ".model.1", x)
    x = res_block(sd, p + ".model.2", x)
    x = se_layer(sd, p + ".model.3", x)
    return rm_conv_block(sd, p + ".model.4", x, norm=False)


def layer_norm_cf(sd, p, x, eps=1e-6):
    """layers/convolutions.py:303-308 channels_first LayerNorm (biased variance, eps inside sqrt)."""
    u = x.mean(1, keepdim=True)
    s = (x - u).pow(2).mean(1, keepdim=True)
    x = (x - u) / torch.sqrt(s + eps)
    return sd[p + ".weight"][:, None, None] * x + sd[p + ".bias"][:, None, None]


def bottleblock(sd, p, x):
    """layers/convolutions.py:348-380 (projection branch: in != out)."""
    r = F.gelu(layer_norm_cf(sd, p + ".layers.1", _conv(sd, p + ".layers.0", x, padding=3)))
    r = F.gelu(layer_norm_cf(sd, p + ".layers.4", _conv(sd, p + ".layers.3", r)))
    r = F.gelu(layer_norm_cf(sd, p + ".layers.7", _conv(sd, p + ".layers.6", r, padding=1)))
    if (p + ".projection.0.weight") in sd:
        return r + F.gelu(_conv(sd, p + ".projection.0", x))
    return r + x


def gru_cell(sd, p, x, state, suffix="", gru_bias_init=0.0):
    """layers/temporal.py:44-57 and layers/temporal_ode_bayes.py:133-161 (no norm/act on the
    candidate; ``suffix`` is ``_1``/``_2`` for the dual cells)."""
    xs = torch.cat([x, state], dim=1)
    u = torch.sigmoid(_conv(sd, f"{p}.conv_update{suffix}", xs, padding=1) + gru_bias_init)
    r = torch.sigmoid(_conv(sd, f"{p}.conv_reset{suffix}", xs, padding=1) + gru_bias_init)
    cand = _conv(sd, f"{p}.conv_state_tilde{suffix}", torch.cat([x, (1.0 - r) * state], dim=1), padding=1)
    return (1.0 - u) * state + u * cand


def dual_cell(sd, p, x, state, derivative, gru_bias_init=0.0):
    """layers/temporal_ode_bayes.py:92-131 (``DualGRUODECell``, derivative=True: returns cur - s)
    and :239-275 (``DualGRUCell``, derivative=False: returns cur).  4-D inputs, n_present = 1."""
    r1 = gru_cell(sd, p, x, state, "_1", gru_bias_init)
    h2 = gru_cell(sd, p, state, state, "_2", gru_bias_init)
    r2 = _conv(sd, p + ".conv_decoder_2", h2, padding=1)
    t = bottleblock(sd, p + ".trusting_gate.0", torch.cat([r1, r2], dim=1))
    g = torch.softmax(_conv(sd, p + ".trusting_gate.1", t), dim=1)
    cur = r2 * g[:, 0:1] + r1 * g[:, 1:]
    return cur - state if derivative else cur


def rsample_normal(raw, eps):
    """models/model_utils.py:60-86,89-109: loc + eps * (softplus(raw_scale) + 1e-8)."""
    loc, raw_scale = torch.chunk(raw, 2, 1)
    return loc + eps * (F.softplus(raw_scale) + 1e-8)


def infer_state(sd, p, x, eps_fn):
    """layers/temporal_ode_bayes.py:463-477."""
    q = conv_net(sd, p + ".p_model", x)
    eps = eps_fn((x.shape[0], q.shape[1] // 2) + tuple(x.shape[2:]), x.dtype, x.device)
    return rsample_normal(q, eps), q


def ode_step(sd, p, state, inp, delta_t, solver, impute, eps_fn):
    """layers/temporal_ode_bayes.py:436-461.  ``delta_t`` is a python float or 0-d float64
    tensor; the multiply happens in fp32 exactly as in the reference."""
    if impute is False:
        inp = torch.zeros_like(inp)
    if solver == "euler":
        state = state + delta_t * dual_cell(sd, p + ".gru_c", inp, state, True)
        inp = infer_state(sd, p, state, eps_fn)[0]
    elif solver == "midpoint":
        k = state + delta_t / 2 * dual_cell(sd, p + ".gru_c", inp, state, True)
        pk = infer_state(sd, p, k, eps_fn)[0]
        state = state + delta_t * dual_cell(sd, p + ".gru_c", pk, k, True)
        inp = infer_state(sd, p, state, eps_fn)[0]
    elif solver == "rk4":
        # BUILD-DEFINED (no reference implementation, SURVEY.md §8c "RK4"): classical tableau
        # composed from the reference's own callables, each stage's imputed input coming from
        # infer_state of the stage state, mirroring how 'midpoint' does it at :450-455.
        f = lambda pp, ss: dual_cell(sd, p + ".gru_c", pp, ss, True)
        k1 = f(inp, state)
        s2 = state + delta_t / 2 * k1
        k2 = f(infer_state(sd, p, s2, eps_fn)[0], s2)
        s3 = state + delta_t / 2 * k2
        k3 = f(infer_state(sd, p, s3, eps_fn)[0], s3)
        s4 = state + delta_t * k3
        k4 = f(infer_state(sd, p, s4, eps_fn)[0], s4)
        state = state + delta_t / 6 * (k1 + 2 * k2 + 2 * k3 + k4)
        inp = infer_state(sd, p, state, eps_fn)[0]
    else:
        raise ValueError(solver)
    return state, inp


def nnfo_forward(sd, p, times, inp, obs, delta_t, T, solver="euler", impute=True, variable=True,
                 eps_fn=None, trace=None):
    """layers/temporal_ode_bayes.py:479-627 ``NNFOwithBayesianJumps.forward``.

    ``times`` 1-D float64 tensor (sorted), ``inp`` (1,1,C,H,W), ``obs`` (1,n_obs,C,H,W),
    ``T`` 1-D float64 tensor.  Returns (state, 0, x) with x (1,len(T),C,H,W).
    ``trace`` (optional list) receives ('step', dt) / ('jump', t) / ('select', idx) entries.
    """
    b, n, c, H, W = obs.shape
    hx_obs = small_encoder(sd, p + ".srvp_encoder", obs.reshape(b * n, c, H, W))
    hx_obs = hx_obs.view(b, n, *hx_obs.shape[1:])                                   # :502
    inp = small_encoder(sd, p + ".srvp_encoder", inp.reshape(-1, c, H, W))          # :503-505
    state = torch.zeros_like(inp)                                                   # :507
    current_time = times.min().item()                                               # :508
    path_t, path_h = [], []

    def step(state, inp, dt, current_time):
        state, inp = ode_step(sd, p, state, inp, dt, solver, impute, eps_fn)
        current_time = current_time + dt                                            # :458
        if isinstance(current_time, torch.Tensor):
            current_time = current_time.item()                                      # :552,:598
        if trace is not None:
            trace.append(("step", float(dt)))
        return state, inp, current_time

    for i, obs_time in enumerate(times):                                            # :539
        while current_time <= (obs_time - delta_t):                                 # :541
            dt = (obs_time - current_time) if variable else delta_t                 # :546-549
            state, inp, current_time = step(state, inp, dt, current_time)
        state = dual_cell(sd, p + ".gru_obs.gru_d", hx_obs[:, i], state, False)     # :562-565
        inp = infer_state(sd, p, state, eps_fn)[0]                                  # :574
        path_t.append(obs_time.item())
        path_h.append(state)                                                        # :578-581
        if trace is not None:
            trace.append(("jump", obs_time.item()))

    for predict_time in T:                                                          # :585
        while current_time < predict_time:                                          # :586
            dt = (predict_time - current_time) if variable else delta_t             # :590-593
            state, inp, current_time = step(state, inp, dt, current_time)
            if current_time > predict_time - 0.5 * delta_t and current_time < predict_time + 0.5 * delta_t:
                path_t.append(current_time)
                path_h.append(state)                                                # :601-604

    xs = []
    path_t = np.array(path_t)
    for time_stamp in T:                                                            # :610-620
        time_stamp = time_stamp.item()
        A = np.where(path_t > time_stamp - 0.5 * delta_t)[0]
        B = np.where(path_t < time_stamp + 0.5 * delta_t)[0]
        if np.any(np.isin(A, B)):
            idx = np.max(A[np.isin(A, B)])
        else:
            idx = np.argmin(np.abs(path_t - time_stamp))
        if trace is not None:
            trace.append(("select", int(idx)))
        xs.append(path_h[idx])
    x = torch.stack(xs, dim=1)                                                      # :622
    bb, t = x.shape[:2]
    x = small_decoder(sd, p + ".srvp_decoder", x.reshape(bb * t, *x.shape[2:]))     # :396-410
    return state, 0, x.view(bb, t, *x.shape[1:])


def spatial_gru(sd, p, x, state=None):
    """layers/temporal.py:26-42."""
    b, T, c, h, w = x.shape
    hid = sd[p + ".conv_update.weight"].shape[0]
    s = torch.zeros(b, hid, h, w) if state is None else state
    out = []
    for t in range(T):
        s = gru_cell(sd, p, x[:, t], s)
        out.append(_conv(sd, p + ".conv_decoder", s))
    return torch.stack(out, dim=1)


def convnext_block(sd, p, x):
    """layers/convolutions.py:333-346 (drop_path = Identity)."""
    C = x.shape[1]
    y = _conv(sd, p + ".dwconv", x, padding=3, groups=C).permute(0, 2, 3, 1)
    y = F.layer_norm(y, (C,), sd[p + ".norm.weight"], sd[p + ".norm.bias"], 1e-6)
    y = F.linear(F.gelu(F.linear(y, sd[p + ".pwconv1.weight"], sd[p + ".pwconv1.bias"])),
                 sd[p + ".pwconv2.weight"], sd[p + ".pwconv2.bias"])
    if (p + ".gamma") in sd:
        y = sd[p + ".gamma"] * y
    return x + y.permute(0, 3, 1, 2)


def deeplab_head(sd, p, x):
    """layers/convolutions.py:217-280 ``DeepLabHead(in, num_classes, hidden)``: ASPP(12,24,36)
    -> 3x3+BN+ReLU -> 1x1 (Dropout inert)."""
    a = p + ".0"
    res = [F.relu(_bn(sd, a + ".convs.0.1", _conv(sd, a + ".convs.0.0", x)))]
    for i, rate in enumerate((12, 24, 36)):
        q = f"{a}.convs.{i + 1}"
        res.append(F.relu(_bn(sd, q + ".1", _conv(sd, q + ".0", x, padding=rate, dilation=rate))))
    q = a + ".convs.4"
    g = F.relu(_bn(sd, q + ".2", _conv(sd, q + ".1", x.mean(dim=(2, 3), keepdim=True))))
    res.append(F.interpolate(g, size=x.shape[-2:], mode="bilinear", align_corners=False))
    y = F.relu(_bn(sd, a + ".project.1", _conv(sd, a + ".project.0", torch.cat(res, dim=1))))
    y = F.relu(_bn(sd, p + ".2", _conv(sd, p + ".1", y, padding=1)))
    return _conv(sd, p + ".4", y)


def merge_observations(camera_states, lidar_states, camera_timestamp, lidar_timestamp, bs):
    """models/future_prediction_ode.py:37-49: dict keyed by 0-d tensors (hash by id, so equal
    times are NOT merged), stable sort by time => camera before lidar on ties."""
    items = []
    if camera_states is not None:
        for i in range(camera_timestamp.shape[1]):
            items.append((camera_timestamp[bs, i].item(), camera_states[bs, i][None]))
    if lidar_states is not None:
        for i in range(lidar_timestamp.shape[1]):
            items.append((lidar_timestamp[bs, i].item(), lidar_states[bs, i][None]))
    items.sort(key=lambda kv: kv[0])
    times = torch.tensor([k for k, _ in items], dtype=torch.float64)
    return times, torch.stack([v for _, v in items], dim=1)


def future_prediction_ode_forward(sd, x_in, camera_states, lidar_states, camera_timestamp,
                                  lidar_timestamp, target_timestamp, delta_t=0.05, n_gru_blocks=2,
                                  solver="euler", impute=True, variable=True, eps_fn=None, p=""):
    """models/future_prediction_ode.py:32-64.  ``p`` is the key prefix of the module ('' or
    'model.future_prediction_ode.')."""
    xs = []
    for bs in range(camera_states.shape[0]):
        times, obs = merge_observations(camera_states, lidar_states, camera_timestamp, lidar_timestamp, bs)
        _, _, px = nnfo_forward(sd, p + "gru_ode", times, x_in, obs, delta_t, target_timestamp[bs],
                                solver, impute, variable, eps_fn)
        xs.append(px)
    x = torch.cat(xs, dim=0)
    hidden = x[:, 0]                                                                # :56
    for i in range(n_gru_blocks):
        x = spatial_gru(sd, f"{p}spatial_grus.{i}", x, hidden)                      # :58
        b, s, c, h, w = x.shape
        x = x.reshape(b * s, c, h, w)
        rb = f"{p}res_blocks.{i}"
        if i < n_gru_blocks - 1:                                                    # :23-26
            j = 0
            while f"{rb}.{j}.dwconv.weight" in sd:
                x = convnext_block(sd, f"{rb}.{j}", x)
                j += 1
        else:
            x = deeplab_head(sd, rb, x)
        x = x.view(b, s, c, h, w)
    return x, 0


# --------------------------------------------------------------------------------------------
# secondary: BEVerse-named classes (mmdet3d/models/beverse/models/*.py) and the unused
# streamingflow/models/distributions.py — signature-compat rows a8 / a17 of SURVEY.md §8
# --------------------------------------------------------------------------------------------
def bottleneck(sd, p, x, downsample=False):
    """streamingflow/layers/convolutions.py:164-172 == beverse basic_modules.py:168-178."""
    L = p + ".layers"
    r = F.relu(_bn(sd, L + ".abn_down_project.0", _conv(sd, L + ".conv_down_project", x)))
    r = F.relu(_bn(sd, L + ".abn.0", _conv(sd, L + ".conv", r, padding=1, stride=2 if downsample else 1)))
    r = F.relu(_bn(sd, L + ".abn_up_project.0", _conv(sd, L + ".conv_up_project", r)))
    if (p + ".projection.conv_skip_proj.weight") in sd:
        if downsample:
            x = F.pad(x, (0, x.shape[-1] % 2, 0, x.shape[-2] % 2), value=0)
            x = F.max_pool2d(x, 2, 2)
        return r + _bn(sd, p + ".projection.bn_skip_proj", _conv(sd, p + ".projection.conv_skip_proj", x))
    return r + x


def beverse_spatial_gru(sd, p, x, state):
    """beverse basic_modules.py:241-284 (flow=None): candidate = conv+BN+ReLU, returns the states."""
    b, T, c, h, w = x.shape
    out = []
    for t in range(T):
        xs = torch.cat([x[:, t], state], dim=1)
        u = torch.sigmoid(_conv(sd, p + ".conv_update", xs, padding=1))
        r = torch.sigmoid(_conv(sd, p + ".conv_reset", xs, padding=1))
        cand = torch.cat([x[:, t], (1.0 - r) * state], dim=1)
        cand = F.relu(_bn(sd, p + ".conv_state_tilde.norm", _conv(sd, p + ".conv_state_tilde.conv", cand, padding=1)))
        state = (1.0 - u) * state + u * cand
        out.append(state)
    return torch.stack(out, dim=1)


def beverse_future_prediction(sd, x, hidden, n_gru_blocks=3, n_res_layers=3, p=""):
    """beverse motion_modules.py:132-146."""
    for i in range(n_gru_blocks):
        x = beverse_spatial_gru(sd, f"{p}spatial_grus.{i}", x, hidden)
        b, n, c, h, w = x.shape
        y = x.reshape(b * n, c, h, w)
        for j in range(n_res_layers):
            y = bottleneck(sd, f"{p}res_blocks.{i}.{j}", y)
        x = y.view(b, n, c, h, w)
    return x


def _dist_encoder(sd, p, x, n):
    for i in range(n):
        x = bottleneck(sd, f"{p}.model.{i}", x, downsample=True)
    return x


def beverse_spatial_distribution(sd, s_t, latent_dim, lo, hi, p=""):
    """beverse motion_modules.py:74-88."""
    e = _dist_encoder(sd, p + "encoder", s_t[:, 0], 2)
    o = _conv(sd, p + "last_conv.0", e)
    return o[:, :latent_dim], torch.clamp(o[:, latent_dim:], lo, hi)


def beverse_distribution(sd, s_t, latent_dim, lo, hi, p=""):
    """beverse motion_modules.py:34-46."""
    b = s_t.shape[0]
    e = _dist_encoder(sd, p + "encoder", s_t[:, 0], 2)
    o = _conv(sd, p + "last_conv.1", e.mean(dim=(2, 3), keepdim=True)).view(b, 1, 2 * latent_dim)
    return o[:, :, :latent_dim], torch.clamp(o[:, :, latent_dim:], lo, hi)


def sf_distribution(sd, s_t, latent_dim, p="", method="GAUSSIAN"):
    """streamingflow/models/distributions.py:35-51: GAUSSIAN / MIXGAUSSIAN (encoder, global average pool, 1x1 conv) or BERNOULLI
    (one Bottleneck + LogSigmoid, :29-33)."""
    b = s_t.shape[0]
    if method == "BERNOULLI":
        return F.logsigmoid(bottleneck(sd, p + "encoder.0", s_t[:, 0], downsample=False))
    e = _dist_encoder(sd, p + "encoder", s_t[:, 0], 4)
    n_out = 2 * latent_dim if method == "GAUSSIAN" else 6 * latent_dim + 3
    return _conv(sd, p + "decoder.1", e.mean(dim=(2, 3), keepdim=True)).view(b, 1, n_out)


def single_gru_cell(sd, x, state, ode, p=""):
    """temporal_ode_bayes.py:35-61 (``SpatialGRUODECell``, ode=True: u*(h~ - s)) and :184-208
    (``SpatialGRUCell``): candidate = ConvBlock (conv, BatchNorm, ReLU)."""
    xs = torch.cat([x, state], dim=1)
    u = torch.sigmoid(_conv(sd, p + "conv_update", xs, padding=1))
    r = torch.sigmoid(_conv(sd, p + "conv_reset", xs, padding=1))
    cand = torch.cat([x, (1.0 - r) * state], dim=1)
    cand = F.relu(_bn(sd, p + "conv_state_tilde.norm", _conv(sd, p + "conv_state_tilde.conv", cand, padding=1)))
    return u * (cand - state) if ode else (1.0 - u) * state + u * cand


# ---- modules the reference defines but never constructs ------------------------------------------------

def _dual_branches(sd, p, x, r1, r2, hid, gru_bias_init=0.0):
    """One pass of the two branches + trusting gate shared by Dual_GRU (layers/temporal.py:112-122) and the dual cells
    (temporal_ode_bayes.py:116-129): returns (cur, new hidden state of branch 2)."""
    r1 = gru_cell(sd, p, x, r1, "_1", gru_bias_init)
    hid = gru_cell(sd, p, r2, hid, "_2", gru_bias_init)
    r2 = _conv(sd, p + ".conv_decoder_2", hid, padding=1)
    t = bottleblock(sd, p + ".trusting_gate.0", torch.cat([r1, r2], dim=1))
    g = torch.softmax(_conv(sd, p + ".trusting_gate.1", t), dim=1)
    return r2 * g[:, 0:1] + r1 * g[:, 1:], hid


def dual_gru(sd, x, state, n_future, mixture=True, gru_bias_init=0.0):
    """layers/temporal.py:88-126 (``Dual_GRU.forward``) on a bare state_dict: x [b, 1, Cin, h, w],
    state [b, n_present, C, h, w] -> [b, n_future, C, h, w]."""
    m = _prefixed(sd)
    hid = state[:, 0]
    for t in range(state.shape[1] - 1):                  # warm-up of branch 2 over the present frames
        hid = gru_cell(m, "m", state[:, t], hid, "_2", gru_bias_init)
    r1 = r2 = state[:, -1]
    out = []
    for _ in range(n_future):
        r1n = gru_cell(m, "m", x[:, 0], r1, "_1", gru_bias_init)
        hid = gru_cell(m, "m", r2, hid, "_2", gru_bias_init)
        r2n = _conv(m, "m.conv_decoder_2", hid, padding=1)
        t = bottleblock(m, "m.trusting_gate.0", torch.cat([r1n, r2n], dim=1))
        g = torch.softmax(_conv(m, "m.trusting_gate.1", t), dim=1)
        cur = r2n * g[:, 0:1] + r1n * g[:, 1:]
        out.append(cur)
        r1, r2 = (cur, cur) if mixture else (r1n, r2n)
    return torch.stack(out, dim=1)


def _prefixed(sd, name="m"):
    """The functions above address parameters as '<prefix>.<key>'; a bare module state_dict gets a prefix."""
    return {name + "." + k: v for k, v in sd.items()}


def bigru(sd, x, gru_bias_init=0.0):
    """layers/temporal.py:183-213 (``BiGRU.forward``) on a bare state_dict: x [b, s, C, h, w]."""
    m = _prefixed(sd)
    b, s, c, h, w = x.shape
    r1, r2 = x[:, 0], x[:, -1]
    fwd, bwd = [], []
    for t in range(s):
        r1 = gru_cell(m, "m", x[:, t], r1, "_1", gru_bias_init)
        r2 = gru_cell(m, "m", x[:, s - t - 1], r2, "_2", gru_bias_init)
        fwd.append(bottleblock(m, "m.conv_decoder_1", r1))
        bwd.append(bottleblock(m, "m.conv_decoder_2", r2))
    states = torch.cat([torch.stack(fwd, dim=1), torch.stack(bwd[::-1], dim=1)], dim=2).view(b * s, 2 * c, h, w)
    y = bottleblock(m, "m.res_blocks.0", states)
    y = convnext_block(m, "m.res_blocks.2", convnext_block(m, "m.res_blocks.1", y))
    return y.view(b, s, c, h, w)


def dual_cell_frames(sd, x, state, derivative, gru_bias_init=0.0):
    """The dual cells with several present frames (temporal_ode_bayes.py:101-131 / :248-275) on a bare state_dict:
    x [b, 1, C, h, w], state [b, n, C, h, w].  Only the ODE cell warms branch 2 up; its result is ``cur - state`` broadcast
    over the frames (the reference's ``squeeze(1)`` is a no-op for n > 1)."""
    m = _prefixed(sd)
    hid = state[:, 0]
    if derivative:
        for t in range(state.shape[1] - 1):
            hid = gru_cell(m, "m", state[:, t], hid, "_2", gru_bias_init)
    cur, _ = _dual_branches(m, "m", x[:, 0], state[:, -1], state[:, -1], hid, gru_bias_init)
    return cur - state if derivative else cur
