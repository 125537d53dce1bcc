"""GPU: the opt-in "bf16x3" math mode (split-bf16 operands on v_mfma_f32_16x16x32_bf16, fp32 accumulators).
Never the default: every other test and the bench headline run exact fp32.  Checked here: the mode is live (results differ
from the exact path), stays within the exact path's own tolerance on every LDS-DMA layer shape, reproduces bit for bit, and
holds the north-star tolerance (1e-3 max-abs on the BEV logits) on a full-size forward with a wide margin."""
import pytest
import torch

import test_gpu_conv_random as RC
from util import build_pair, cases, hashfill, maxabs

pytestmark = pytest.mark.gpu


@pytest.fixture
def bf16x3():
    import streamingflow_amd as sfa
    sfa.set_math_mode("bf16x3")
    yield
    sfa.set_math_mode("fp32")


@pytest.mark.parametrize("i", range(len(RC._DMA)))
def test_dma_layers_in_bf16x3(i, bf16x3):
    import streamingflow_amd as sfa
    c = dict(act=["none", "relu", "lrelu", "tanh"][i % 4], add=i % 3 != 0, after=i % 2 == 0, in_slack=8 * (i % 2), out_slack=[0, 4, 16][i % 3])
    c.update(RC._DMA[i])
    got3 = RC._run(c, 100 + i)                 # the exact path's own tolerance (2e-4 against torch CPU fp32)
    again = RC._run(c, 100 + i)
    assert torch.equal(got3, again)            # fixed summation order here too
    sfa.set_math_mode("fp32")
    got = RC._run(c, 100 + i)
    assert not torch.equal(got3, got), "bf16x3 mode was not used"
    assert maxabs(got3, got) <= 5e-5


def test_full_size_forward_in_bf16x3(bf16x3):
    """BASELINE config 2 (C=64, BEV 200x200, 10 Euler steps + 8 jumps) with every LDS-DMA layer in bf16x3 against the oracle."""
    from oracle import ref_torch as R
    C, H, W = 64, 200, 200
    cts, lts, tts, dt = cases.timeset("shipped")
    net, sd = build_pair(C, "euler", True, True, dt)
    cam, lid = cases.bev_inputs(C, H, W, cts.shape[1], lts.shape[1])
    net.gru_ode.noise = hashfill.HashedNoise(cases.EPS_SEED)
    y, _ = net(cases.present_input(cam, lid).cuda(), cam.cuda(), lid.cuda(), cts, lts, tts)
    torch.set_num_threads(16)
    with torch.no_grad():
        yr, _ = R.future_prediction_ode_forward(sd, cases.present_input(cam, lid), cam, lid, cts, lts, tts, dt, 2, "euler", True, True,
                                                hashfill.HashedNoise(cases.EPS_SEED))
    assert bool(torch.isfinite(y).all())      # the mode is for finite activations (an Inf operand would give NaN)
    err = maxabs(y, yr)
    print("bf16x3 full-size forward max-abs vs oracle", err)
    assert err <= 2e-4      # north-star tolerance 1e-3; measured ~5e-5


@pytest.mark.parametrize("solver", ["euler", "rk4"])
def test_single_latent_stream40_rollout_in_bf16x3(solver):
    """The 46-step streaming rollout (BASELINE config 5) on ONE 50x50x64 latent — the small-P kernel's split-bf16 loop, the
    pipelined stages included — against the exact-fp32 rollout of the same inputs (itself checked against the oracle and the
    reference in test_gpu_configs.py): chained over 46 steps (184 cell evaluations for rk4) the states stay within 2e-4."""
    import streamingflow_amd as sfa
    from streamingflow_amd import schedule as S
    C, h, w = 64, 50, 50
    cts, lts, tts, dt = cases.timeset("stream40")
    net, _ = build_pair(C, solver, True, True, dt)
    ode = net.gru_ode
    times, _ = S.merge_observations(cts[0].tolist(), lts[0].tolist())
    sc = S.build_schedule(times, tts[0].tolist(), dt, True, solver)
    hx = (hashfill.normal("b3hx", (8, h, w, C), 61) * 0.5).cuda()
    eps = hashfill.normal("b3eps", (sc.n_draws, h, w, C), 62).cuda()
    a, fa = ode.rollout_nhwc(hx, sc, eps)
    a, fa = a.clone(), fa.clone()
    sfa.set_math_mode("bf16x3")
    try:
        b, fb = ode.rollout_nhwc(hx, sc, eps)
        b2, _ = ode.rollout_nhwc(hx, sc, eps)
        assert torch.equal(b, b2)
    finally:
        sfa.set_math_mode("fp32")
    assert not torch.equal(a, b), "bf16x3 mode was not used"
    err = max(maxabs(a, b), maxabs(fa, fb))
    print(f"stream40 {solver}: bf16x3 vs fp32 max-abs over the 43 selected states", err, "scale", float(a.abs().max()))
    assert err <= 2e-4


def test_bn_fold_is_the_librarys_own():
    """packing.bn_fold is a caller of sf_bn_fold (the device code sf_pack_conv folds with): compare with the textbook formula."""
    import torch.nn as nn
    from streamingflow_amd import packing
    bn = nn.BatchNorm2d(20, eps=1e-3).cuda().eval()
    with torch.no_grad():
        bn.weight.copy_(hashfill.uniform("bnw", (20,), 0.5, 1.5, 7)); bn.bias.copy_(hashfill.uniform("bnb", (20,), -1, 1, 8))
        bn.running_mean.copy_(hashfill.uniform("bnm", (20,), -1, 1, 9)); bn.running_var.copy_(hashfill.uniform("bnv", (20,), 0.2, 2.0, 10))
    cb = hashfill.uniform("bncb", (20,), -1, 1, 11).cuda()
    sc, bi = packing.bn_fold(bn, cb)
    want_sc = bn.weight / torch.sqrt(bn.running_var + bn.eps)
    want_bi = bn.bias - bn.running_mean * want_sc + cb * want_sc
    assert maxabs(sc, want_sc) <= 1e-6 and maxabs(bi, want_bi) <= 2e-6
    sc2, bi2 = packing.bn_fold(bn)
    assert maxabs(bi2, bn.bias - bn.running_mean * want_sc) <= 2e-6


def test_math_mode_switch_repacks_and_drops_graphs():
    """set_math_mode changes the pack signature: modules re-pack on their next call and captured rollout graphs (raw pointers
    into the old packs) are dropped; switching back restores the exact results bit for bit."""
    import streamingflow_amd as sfa
    from streamingflow_amd import schedule as S
    C, h, w = 16, 12, 12
    cts, lts, tts, dt = cases.timeset("shipped")
    net, _ = build_pair(C, "euler", True, True, dt)
    ode = net.gru_ode
    times, _ = S.merge_observations(cts[0].tolist(), lts[0].tolist())
    sc = S.build_schedule(times, tts[0].tolist(), dt, True, "euler")
    hx = (hashfill.normal("mmhx", (len(times), h, w, C), 61) * 0.5).cuda()
    eps = hashfill.normal("mmeps", (sc.n_draws, h, w, C), 62).cuda()
    ode.use_graph = True
    a, _ = ode.rollout_nhwc(hx, sc, eps); a = a.clone()
    g0 = ode.gru_c.pack_generation()
    sfa.set_math_mode("bf16x3")
    try:
        b, _ = ode.rollout_nhwc(hx, sc, eps); b = b.clone()
        assert ode.gru_c.pack_generation() != g0 and len(ode._graphs) == 1
    finally:
        sfa.set_math_mode("fp32")
    c, _ = ode.rollout_nhwc(hx, sc, eps)
    assert torch.equal(a, c) and maxabs(a, b) <= 2e-4
    ode.use_graph = False


@pytest.mark.parametrize("tag", ["config4_future16", "config1_c32", "config5_stream40_euler", "config5_stream40_midpoint"])
def test_full_size_configs_in_bf16x3_vs_reference_statistics(tag, bf16x3):
    """The mode at full size on the other BASELINE configurations (1: C=32, 4 fixed Euler steps; 4: 19 frames, 8 s horizon; 5: the
    46-step streaming schedule, 43 frames, euler and midpoint) against the statistics + 256 samples of the REAL reference
    (tests/golden/big_stats.json): within the north-star tolerance, with the measured margin printed."""
    import json
    import os
    import test_gpu_configs as TC
    cfgs = {**cases.BIG_CASES, **cases.BIG_STREAM_CASES}
    C, H, W, ts, solver, impute, variable = cfgs[tag]
    y, _, _ = TC._forward(C, H, W, ts, solver, impute, variable)
    st = json.load(open(os.path.join(TC.GOLD, "big_stats.json")))["cases"][tag]["out"]
    flat = y.reshape(-1).double().cpu()
    err = float((flat[torch.tensor(st["sample_idx"])] - torch.tensor(st["samples"])).abs().max())
    print(f"bf16x3 {tag}: max-abs over the reference's 256 samples {err:.2e}, |mean - ref| {abs(flat.mean().item() - st['mean']):.1e}")
    assert list(y.shape) == st["shape"] and err <= 1e-3 and abs(flat.mean().item() - st["mean"]) <= 1e-4
