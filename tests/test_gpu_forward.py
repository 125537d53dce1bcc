"""GPU parity, end to end: FuturePredictionODE.forward on libsfnative against the fixtures
generated from the real reference (tests/golden/fpode.npz, big_stats.json) and against the oracle
at BASELINE config 2's full size.  North-star tolerance: <= 1e-3 max-abs on the fp32 BEV output."""
import json
import os

import numpy as np
import pytest
import torch

from util import GOLD, cases, hashfill, gold, maxabs, build_pair
from oracle import ref_torch as R

pytestmark = pytest.mark.gpu
TOL_E2E = 1e-3


def _run(name):
    C, H, W, ts, solver, impute, variable, eps0 = cases.FPODE_CASES[name]
    cts, lts, tts, dt = cases.timeset(ts)
    net, sd = build_pair(C, solver, impute, variable, dt)
    cam, lid = cases.bev_inputs(C, H, W, cts.shape[1], lts.shape[1])
    net.gru_ode.noise = hashfill.HashedNoise(cases.EPS_SEED, zero=eps0)
    y, aux = net(cases.present_input(cam, lid).cuda(), cam.cuda(), lid.cuda(), cts, lts, tts)
    return y, aux


@pytest.mark.parametrize("name", list(cases.FPODE_CASES))
def test_fpode_golden(name):
    y, aux = _run(name)
    assert aux == 0
    g = gold("fpode.npz")
    assert maxabs(y, g[name + "/out"]) <= TOL_E2E


def test_nnfo_forward_golden():
    """NNFOwithBayesianJumps.forward alone (state + decoded predictions)."""
    name = "c8_16_shipped"
    C, H, W, ts, solver, impute, variable, eps0 = cases.FPODE_CASES[name]
    cts, lts, tts, dt = cases.timeset(ts)
    net, sd = build_pair(C, solver, impute, variable, dt)
    cam, lid = cases.bev_inputs(C, H, W, cts.shape[1], lts.shape[1])
    times, obs = R.merge_observations(cam, lid, cts, lts, 0)
    net.gru_ode.noise = hashfill.HashedNoise(cases.EPS_SEED)
    state, loss, x = net.gru_ode(times, cases.present_input(cam, lid).cuda(), obs.cuda(), dt, tts[0])
    g = gold("fpode.npz")
    assert loss == 0
    assert maxabs(state, g[name + "/nnfo_state"]) <= 1e-4
    assert maxabs(x, g[name + "/nnfo_x"]) <= TOL_E2E


def test_config2_full_size_vs_oracle_and_reference_stats():
    """BASELINE config 2: C=64, BEV 200x200, 3 camera + 5 LiDAR observations, 7 targets, variable
    step (10 steps + 8 jumps).  Checked against the oracle (same process, host cores) and against
    the statistics of the real reference's output (tests/golden/big_stats.json)."""
    C, H, W = 64, 200, 200
    cts, lts, tts, dt = cases.timeset("shipped")
    net, sd = build_pair(C, "euler", True, True, dt)
    cam, lid = cases.bev_inputs(C, H, W, 3, 5)
    net.gru_ode.noise = hashfill.HashedNoise(cases.EPS_SEED)
    y, _ = net(cases.present_input(cam, lid).cuda(), cam.cuda(), lid.cuda(), cts, lts, tts)
    y = y.cpu()
    assert y.shape == (1, 7, C, H, W)
    path = os.path.join(GOLD, "big_stats.json")
    if os.path.exists(path):
        st = json.load(open(path))["out"]
        flat = y.reshape(-1).double()
        samples = flat[torch.tensor(st["sample_idx"])]
        assert float((samples - torch.tensor(st["samples"])).abs().max()) <= TOL_E2E
        assert abs(flat.mean().item() - st["mean"]) <= 1e-4
        assert abs(flat.abs().max().item() - st["absmax"]) <= TOL_E2E
    torch.set_num_threads(min(os.cpu_count() or 8, 16))
    with torch.no_grad():
        yr, _ = R.future_prediction_ode_forward(sd, cases.present_input(cam, lid), cam, lid, cts, lts, tts, dt, 2,
                                                "euler", True, True, hashfill.HashedNoise(cases.EPS_SEED))
    assert maxabs(y, yr) <= TOL_E2E


class _SameNoise:
    """k-th draw: one hashed sample, repeated over the batch (any batch size sees the same per-sample noise)."""

    def __init__(self):
        self.k = 0

    def __call__(self, shape, dtype=torch.float32, device="cpu"):
        e = hashfill.normal(f"same_eps{self.k}", (1,) + tuple(shape[1:]), 7)
        self.k += 1
        return e.expand(shape[0], *e.shape[1:]).contiguous()


def test_config2_batch32_equals_single_sample():
    """The bench workload (32 samples per forward: 224-frame tensors of up to 4.6 GB, tiles that straddle images, the
    large-tile LDS-DMA kernels with 32-bit rebased offsets) against the single-sample forward (small-P kernels) on 32
    copies of one sample: every copy must equal the single-sample result to 1e-4 and the copies must agree with each
    other (same kernels, different tile positions and addresses)."""
    C, H, W, B = 64, 200, 200, 32
    cts, lts, tts, dt = cases.timeset("shipped")
    net, _ = build_pair(C, "euler", True, True, dt)
    cam, lid = cases.bev_inputs(C, H, W, 3, 5)
    cam, lid = cam.cuda(), lid.cuda()
    pres = cases.present_input(cam, lid)
    net.gru_ode.noise = _SameNoise()
    y1, _ = net(pres, cam, lid, cts, lts, tts)
    rep = lambda t: t.expand(B, *t.shape[1:]).contiguous()
    net.gru_ode.noise = _SameNoise()
    yb, _ = net(rep(pres), rep(cam), rep(lid), cts.expand(B, -1).contiguous(), lts.expand(B, -1).contiguous(),
                tts.expand(B, -1).contiguous())
    assert yb.shape == (B,) + tuple(y1.shape[1:])
    assert float((yb - yb[:1]).abs().max()) <= 1e-6
    assert float((yb - y1).abs().max()) <= 1e-4


def test_rollout_properties_long_horizon():
    """Size-independent properties at full size (46-step streaming schedule, C=64, 50x50 latent):
    (1) the rollout is deterministic for fixed eps (bitwise: fixed-order reductions, no atomics);
    (2) targets answered by the same visited state are bitwise equal;
    (3) with IMPUTE off the result does not depend on eps at all."""
    from streamingflow_amd import schedule as S
    C, h, w = 64, 50, 50
    cts, lts, tts, dt = cases.timeset("stream40")
    net, _ = build_pair(C, "euler", True, True, dt)
    ode = net.gru_ode
    times, _ = S.merge_observations(cts[0].tolist(), lts[0].tolist())
    sc = S.build_schedule(times, tts[0].tolist() + [tts[0, -1].item()], dt, True)
    hx = (hashfill.normal("hx", (8, h, w, C), 41) * 0.5).cuda()
    eps = hashfill.normal("epsL", (sc.n_draws, h, w, C), 42).cuda()
    a, fa = ode.rollout_nhwc(hx, sc, eps)
    b, fb = ode.rollout_nhwc(hx, sc, eps)
    assert torch.equal(a, b) and torch.equal(fa, fb)
    assert torch.equal(a[-1], a[-2]) and torch.equal(a[-1], fa)
    assert torch.isfinite(a).all()
    ode.impute = False
    c, _ = ode.rollout_nhwc(hx, sc, eps)
    d, _ = ode.rollout_nhwc(hx, sc, eps * 0 + 3.0)
    assert torch.equal(c, d)
    assert not torch.equal(a, c)


class _PerSampleNoise:
    """eps source for batched calls: draw k of sample b is hashfill.normal(f'eps{k}_s{b}')."""

    def __init__(self, samples):
        self.samples, self.k = list(samples), 0

    def __call__(self, shape, dtype=torch.float32, device="cpu"):
        k = self.k
        self.k += 1
        assert shape[0] == len(self.samples)
        return torch.cat([hashfill.normal(f"eps{k}_s{b}", (1,) + tuple(shape[1:]), 7) for b in self.samples], 0)


def test_batched_forward_matches_per_sample_oracle():
    """Three samples with the same schedule structure but different step sizes go through the
    encoder / rollout / head as ONE batch (per-image dt coefficients, per-image SE gates, per-image
    ASPP pooling); each must equal the oracle run on that sample alone."""
    C, H, W = 16, 32, 32
    cam_ts = torch.tensor([[-1.0, -.5, 0], [-.97, -.52, -.01], [-1.02, -.49, 0]], dtype=torch.float64)
    lid_ts = torch.tensor([[-.8, -.6, -.4, -.2, 0], [-.83, -.61, -.42, -.17, -.01], [-.79, -.6, -.38, -.2, 0]], dtype=torch.float64)
    tgt_ts = torch.tensor([[.5, 1, 1.5, 2], [.52, 1.01, 1.49, 2.0], [.5, .98, 1.5, 2.03]], dtype=torch.float64)
    net, sd = build_pair(C, "euler", True, True, 0.05)
    cam = hashfill.normal("bcam", (3, 3, C, H, W), 9)
    lid = hashfill.normal("blid", (3, 5, C, H, W), 10)
    net.gru_ode.noise = _PerSampleNoise([0, 1, 2])
    y, aux = net(cam[:, -1:].cuda(), cam.cuda(), lid.cuda(), cam_ts, lid_ts, tgt_ts)
    assert aux == 0 and y.shape == (3, 4, C, H, W)
    for b in range(3):
        k = [0]

        def eps_fn(shape, dtype, device, b=b, k=k):
            k[0] += 1
            return hashfill.normal(f"eps{k[0] - 1}_s{b}", tuple(shape), 7)

        with torch.no_grad():
            yr, _ = R.future_prediction_ode_forward(sd, cam[b:b + 1, -1:], cam[b:b + 1], lid[b:b + 1], cam_ts[b:b + 1],
                                                    lid_ts[b:b + 1], tgt_ts[b:b + 1], 0.05, 2, "euler", True, True, eps_fn)
        assert maxabs(y[b:b + 1], yr) <= TOL_E2E, b


def test_mixed_schedule_structures_in_one_batch():
    """Samples with different schedule structures are split into groups transparently."""
    C, H, W = 8, 16, 16
    cam_ts = torch.tensor([[-1.0, -.5, 0], [-1.0, -.5, 0]], dtype=torch.float64)
    lid_ts = torch.tensor([[-.8, -.6, -.4, -.2, 0], [-.8, -.6, -.45, -.2, 0]], dtype=torch.float64)   # -.45: extra step
    tgt_ts = torch.tensor([[.5, 1.0], [.5, 1.0]], dtype=torch.float64)
    net, sd = build_pair(C, "euler", True, True, 0.05)
    cam = hashfill.normal("mcam", (2, 3, C, H, W), 9)
    lid = hashfill.normal("mlid", (2, 5, C, H, W), 10)
    net.gru_ode.noise = hashfill.HashedNoise(0, zero=True)
    y, _ = net(cam[:, -1:].cuda(), cam.cuda(), lid.cuda(), cam_ts, lid_ts, tgt_ts)
    for b in range(2):
        with torch.no_grad():
            yr, _ = R.future_prediction_ode_forward(sd, cam[b:b + 1, -1:], cam[b:b + 1], lid[b:b + 1], cam_ts[b:b + 1],
                                                    lid_ts[b:b + 1], tgt_ts[b:b + 1], 0.05, 2, "euler", True, True,
                                                    hashfill.HashedNoise(0, zero=True))
        assert maxabs(y[b:b + 1], yr) <= TOL_E2E, b


def test_hipgraph_rollout_replay_matches_eager():
    """The captured hipGraph of a rollout gives bitwise the eager result, also when replayed with
    new observations / noise / step sizes of the same schedule structure."""
    from streamingflow_amd import schedule as S
    C, h, w = 64, 50, 50
    cts, lts, tts, dt = cases.timeset("shipped")
    net, _ = build_pair(C, "euler", True, True, dt)
    ode = net.gru_ode
    times, _ = S.merge_observations(cts[0].tolist(), lts[0].tolist())
    sc1 = S.build_schedule(times, tts[0].tolist(), dt, True)
    sc2 = S.build_schedule([t * 1.01 for t in times], [t * 1.01 for t in tts[0].tolist()], dt, True)
    assert sc1.key() == sc2.key() and sc1.dts != sc2.dts
    for k, sc in enumerate((sc1, sc2, sc1)):
        hx = (hashfill.normal(f"ghx{k}", (8, h, w, C), 51) * 0.5).cuda()
        eps = hashfill.normal(f"geps{k}", (sc.n_draws, h, w, C), 52).cuda()
        ode.use_graph = False
        a, fa = ode.rollout_nhwc(hx, sc, eps)
        a, fa = a.clone(), fa.clone()
        ode.use_graph = True
        b, fb = ode.rollout_nhwc(hx, sc, eps)
        assert torch.equal(a, b) and torch.equal(fa, fb), k
    assert len(ode._graphs) == 1
    ode.use_graph = False


def test_bev_size_not_divisible_by_four():
    """The reference silently shrinks the output when H, W are not multiples of 4 (two floor
    max-pools, then x4 nearest upsampling: 50x50 BEV -> 12x12 latent -> 48x48 out);
    rectangular 50x38 here.  Same behaviour, same numbers as the oracle."""
    C, H, W = 8, 50, 38
    cts, lts, tts, dt = cases.timeset("camera_only")
    net, sd = build_pair(C, "euler", True, True, dt)
    cam, _ = cases.bev_inputs(C, H, W, cts.shape[1], 0)
    net.gru_ode.noise = hashfill.HashedNoise(cases.EPS_SEED)
    y, _ = net(cam[:, -1:].cuda(), cam.cuda(), None, cts, None, tts)
    with torch.no_grad():
        yr, _ = R.future_prediction_ode_forward(sd, cam[:, -1:], cam, None, cts, None, tts, dt, 2, "euler", True, True,
                                                hashfill.HashedNoise(cases.EPS_SEED))
    assert y.shape == yr.shape == (1, 3, C, 48, 36)
    assert maxabs(y, yr) <= TOL_E2E


def test_module_defaults_graph_and_in_kernel_noise():
    """Module defaults: with no noise source injected a single-sample rollout is replayed from a
    captured hipGraph and eps comes from the sampling epilogue (Philox); the returned tensors are fresh (not the graph's
    static buffers); a batch big enough for the large-tile kernels runs eagerly.  With IMPUTE off the result does not
    depend on eps, so default mode == eager + torch.randn bitwise."""
    C, H, W = 16, 32, 32
    cts, lts, tts, dt = cases.timeset("shipped")
    net, _ = build_pair(C, "euler", False, True, dt)
    ode = net.gru_ode
    assert ode.use_graph is None and ode.in_kernel_noise is None and ode.noise is None
    cam, lid = cases.bev_inputs(C, H, W, 3, 5)
    args = (cases.present_input(cam, lid).cuda(), cam.cuda(), lid.cuda(), cts, lts, tts)
    y1, _ = net(*args)
    assert len(ode._graphs) == 1 and ode._noise_calls == 1
    y2, _ = net(*args)
    assert len(ode._graphs) == 1 and ode._noise_calls == 2 and y1.data_ptr() != y2.data_ptr()
    ode.use_graph, ode.in_kernel_noise = False, False
    y3, _ = net(*args)
    assert torch.equal(y1, y2) and torch.equal(y1, y3)
    ode.use_graph, ode.in_kernel_noise = None, None
    # rollout outputs in auto mode are clones: a second replay must not change the first result
    from streamingflow_amd import schedule as S
    times, _ = S.merge_observations(cts[0].tolist(), lts[0].tolist())
    sc = S.build_schedule(times, tts[0].tolist(), dt, True)
    hx = (hashfill.normal("dhx", (8, 8, 8, C), 5) * 0.5).cuda()
    a, fa = ode.rollout_nhwc(hx, sc)
    keep = a.clone()
    b, fb = ode.rollout_nhwc(hx * 2.0, sc)
    assert torch.equal(a, keep) and not torch.equal(a, b)
    # 5000 latent pixels: large-tile kernels, no auto graph
    n_before = len(ode._graphs)
    hx2 = (hashfill.normal("dhx2", (8, 2, 50, 50, C), 6) * 0.5).cuda()
    ode.rollout_nhwc(hx2, sc)
    assert len(ode._graphs) == n_before


def test_in_kernel_noise_follows_torch_seed_and_graph_cache_is_bounded():
    """(a) the default in-kernel (Philox) noise is keyed by torch's global seed, the rank and the module instance — the same
    torch.manual_seed gives the same forward, another seed another one, two modules draw different streams; seed_noise pins it;
    (b) the auto graph cache keeps at most GRAPH_CACHE_MAX captured rollouts (least recently used out) and stops capturing after
    GRAPH_AUTO_MAX_STRUCTURES distinct schedule structures."""
    from streamingflow_amd import schedule as S
    C = 16
    cts, lts, tts, dt = cases.timeset("shipped")
    times, _ = S.merge_observations(cts[0].tolist(), lts[0].tolist())
    sc = S.build_schedule(times, tts[0].tolist(), dt, True)
    hx = (hashfill.normal("seedhx", (8, 8, 8, C), 5) * 0.5).cuda()

    def fresh(seed):
        torch.manual_seed(seed)
        net, _ = build_pair(C, "euler", True, True, dt)
        return net.gru_ode
    o1 = fresh(1234)
    a1, _ = o1.rollout_nhwc(hx, sc)
    serial = o1._noise_serial
    o2 = fresh(1234)
    o2._noise_serial = serial                                   # the same "instance number": same stream
    a2, _ = o2.rollout_nhwc(hx, sc)
    assert torch.equal(a1, a2)
    o3 = fresh(99)
    o3._noise_serial = serial
    a3, _ = o3.rollout_nhwc(hx, sc)
    assert not torch.equal(a1, a3)
    o4 = fresh(1234)                                            # another instance under the same seed: another stream
    a4, _ = o4.rollout_nhwc(hx, sc)
    assert o4._noise_serial != serial and not torch.equal(a1, a4)
    o4.seed_noise(o1.noise_seed)
    a5, _ = o4.rollout_nhwc(hx, sc)
    assert torch.equal(a1, a5)
    # (b) many schedule structures
    ode = o1
    ode.GRAPH_CACHE_MAX, ode.GRAPH_AUTO_MAX_STRUCTURES = 3, 5
    ode.drop_graphs()
    for k in range(8):
        tt = [0.05 * (j + 1) for j in range(k + 1)]
        s_k = S.build_schedule(times, tt, dt, True)
        ode.rollout_nhwc(hx, s_k)
        assert len(ode._graphs) <= 3
    assert len(ode._graphs) == 3 and len(ode._graph_structures_seen) == ode.GRAPH_AUTO_MAX_STRUCTURES + 1      # the set stops growing at the cap
    n = len(ode._graphs)
    ode.rollout_nhwc(hx, S.build_schedule(times, [0.05 * (j + 1) for j in range(12)], dt, True))      # a 9th structure: eager
    assert len(ode._graphs) == n


def test_grad_enabled_inputs_are_refused_parameters_are_not():
    """SURVEY §8(b) / VERDICT r3 item 9b: the HIP path records no autograd history.  Calling outside no_grad() works
    (parameters keep requires_grad=True) and returns detached tensors; an INPUT that asks for gradients raises."""
    C, H, W = 8, 16, 16
    cts, lts, tts, dt = cases.timeset("camera_only")
    net, _ = build_pair(C, "euler", True, True, dt)
    cam, _ = cases.bev_inputs(C, H, W, cts.shape[1], 0)
    x = cam.cuda()
    assert torch.is_grad_enabled() and any(p.requires_grad for p in net.parameters())
    y, _ = net(x[:, -1:], x, None, cts, None, tts)
    assert not y.requires_grad
    xg = x.clone().requires_grad_(True)
    with pytest.raises(RuntimeError, match="inference-only"):
        net(xg[:, -1:], xg, None, cts, None, tts)
    with torch.no_grad():
        y2, _ = net(xg[:, -1:], xg, None, cts, None, tts)
    assert y2.shape == y.shape

