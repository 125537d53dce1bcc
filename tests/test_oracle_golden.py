"""CPU: the oracle (oracle/ref_torch.py) reproduces the fixtures generated from the real reference
(tests/golden, made by oracle/gen_golden.py).  This is what pins the oracle."""
import numpy as np
import pytest
import torch

from util import cases, hashfill, gold, maxabs, build_pair
from oracle import ref_torch as R

TOL = 1e-6


@pytest.fixture(scope="module")
def sd8():
    return build_pair(8, device="cpu")[1]


def test_ops_c8(sd8):
    g = gold("ops_c8.npz")
    C, h, w = 8, 12, 12
    x = hashfill.normal("op_x", (1, C, h, w), 11)
    s = hashfill.normal("op_s", (1, C, h, w), 12) * 0.5
    eps = lambda: hashfill.HashedNoise(cases.EPS_SEED)
    with torch.no_grad():
        assert maxabs(R.gru_cell(sd8, "spatial_grus.0", x, s), g["spatial_gru_cell"]) <= TOL
        assert maxabs(R.dual_cell(sd8, "gru_ode.gru_c", x, s, True), g["dual_ode_cell"]) <= TOL
        assert maxabs(R.dual_cell(sd8, "gru_ode.gru_obs.gru_d", x, s, False), g["dual_obs_cell"]) <= TOL
        y, q = R.infer_state(sd8, "gru_ode", s, eps())
        assert maxabs(y, g["infer_state_y"]) <= TOL and maxabs(q, g["infer_state_q"]) <= TOL
        bev = hashfill.normal("op_bev", (1, 2, C, 4 * h, 4 * w), 13)
        assert maxabs(R.small_encoder(sd8, "gru_ode.srvp_encoder", bev[0]), g["srvp_encode"][0]) <= TOL
        lat = hashfill.normal("op_lat", (1, 2, C, h, w), 14) * 0.5
        assert maxabs(R.small_decoder(sd8, "gru_ode.srvp_decoder", lat[0]), g["srvp_decode"][0]) <= TOL
        frames = hashfill.normal("op_frames", (3, C, 4 * h, 4 * w), 15)
        assert maxabs(R.convnext_block(sd8, "res_blocks.0.0", frames), g["convnext_block"]) <= TOL
        assert maxabs(R.deeplab_head(sd8, "res_blocks.1", frames), g["deeplab_head"]) <= TOL
        seq = hashfill.normal("op_seq", (1, 3, C, 4 * h, 4 * w), 16)
        assert maxabs(R.spatial_gru(sd8, "spatial_grus.1", seq, seq[:, 0]), g["spatial_gru_seq"]) <= TOL
        for solver in ("euler", "midpoint"):
            for impute in (True, False):
                for dt in (0.05, torch.tensor(0.37, dtype=torch.float64)):
                    st, inp = R.ode_step(sd8, "gru_ode", s, x, dt, solver, impute, eps())
                    tag = f"ode_step_{solver}_{'imp' if impute else 'noimp'}_{float(dt):.2f}"
                    assert maxabs(st, g[tag + "_state"]) <= TOL, tag
                    assert maxabs(inp, g[tag + "_input"]) <= TOL, tag


@pytest.mark.parametrize("name", list(cases.FPODE_CASES))
def test_fpode_forward(name):
    C, H, W, ts, solver, impute, variable, eps0 = cases.FPODE_CASES[name]
    if H >= 48 and C > 8:
        pytest.skip("kept small for the CPU suite")
    g = gold("fpode.npz")
    cts, lts, tts, dt = cases.timeset(ts)
    _, sd = build_pair(C, solver, impute, variable, dt, device="cpu")
    cam, lid = cases.bev_inputs(C, H, W, cts.shape[1], lts.shape[1])
    with torch.no_grad():
        y, aux = R.future_prediction_ode_forward(sd, cases.present_input(cam, lid), cam, lid, cts, lts, tts, dt, 2,
                                                 solver, impute, variable, hashfill.HashedNoise(cases.EPS_SEED, zero=eps0))
    assert aux == 0
    assert maxabs(y, g[name + "/out"]) <= 1e-5


@pytest.mark.parametrize("name", list(cases.FPODE_STREAM_CASES))
def test_fpode_stream_forward(name):
    """BASELINE configs 5 (46-step streaming schedule, euler / midpoint) and 4 (19 frames) at toy size: the oracle
    against whole outputs of the real reference (tests/golden/fpode_stream.npz)."""
    C, H, W, ts, solver, impute, variable, eps0 = cases.FPODE_STREAM_CASES[name]
    if C > 8:
        pytest.skip("kept small for the CPU suite")
    g = gold("fpode_stream.npz")
    cts, lts, tts, dt = cases.timeset(ts)
    _, sd = build_pair(C, solver, impute, variable, dt, device="cpu")
    cam, lid = cases.bev_inputs(C, H, W, cts.shape[1], lts.shape[1])
    with torch.no_grad():
        y, aux = R.future_prediction_ode_forward(sd, cases.present_input(cam, lid), cam, lid, cts, lts, tts, dt, 2,
                                                 solver, impute, variable, hashfill.HashedNoise(cases.EPS_SEED, zero=eps0))
    assert aux == 0
    assert maxabs(y, g[name + "/out"]) <= 1e-5


def test_config1_full_size_oracle_vs_reference_stats():
    """BASELINE config 1 exactly as SURVEY.md §8d states it (C=32, BEV 200x200, one camera observation, four fixed Euler
    steps): the oracle against the statistics of the real reference's output (tests/golden/big_stats.json)."""
    import json
    import os
    from util import GOLD
    st = json.load(open(os.path.join(GOLD, "big_stats.json")))["cases"]["config1_c32"]["out"]
    C, H, W, ts, solver, impute, variable = cases.BIG_CASES["config1_c32"]
    cts, lts, tts, dt = cases.timeset(ts)
    _, sd = build_pair(C, solver, impute, variable, dt, device="cpu")
    cam, lid = cases.bev_inputs(C, H, W, cts.shape[1], lts.shape[1])
    with torch.no_grad():
        y, _ = R.future_prediction_ode_forward(sd, cases.present_input(cam, lid), cam, lid, cts, lts, tts, dt, 2,
                                               solver, impute, variable, hashfill.HashedNoise(cases.EPS_SEED))
    assert list(y.shape) == st["shape"]
    flat = y.reshape(-1).double()
    assert float((flat[torch.tensor(st["sample_idx"])] - torch.tensor(st["samples"])).abs().max()) <= 1e-5
    assert abs(flat.mean().item() - st["mean"]) <= 1e-6


def test_gate_bias_cells():
    """gru_bias_init != 0: the oracle against the reference's own cells (tests/golden/gate_bias.npz)."""
    g = gold("gate_bias.npz")
    C, h, w = 8, 12, 12
    x = hashfill.normal("gb_x", (1, C, h, w), 21)
    s = hashfill.normal("gb_s", (1, C, h, w), 22) * 0.5
    import streamingflow_amd as sfa
    from streamingflow_amd.layers import temporal_ode_bayes as tob
    with torch.no_grad():
        for tag, gb in (("pos", 0.7), ("neg", -1.3)):
            sd = {"c." + k: v for k, v in hashfill.fill_state_dict(sfa.layers.temporal.SpatialGRU(C, C).state_dict(), seed=31, gain=0.6).items()}
            assert maxabs(R.gru_cell(sd, "c", x, s, "", gb), g[f"spatial_gru_cell_{tag}"]) <= TOL
            for key, cls, seed, deriv in (("dual_ode_cell", tob.DualGRUODECell, 32, True), ("dual_cell", tob.DualGRUCell, 33, False)):
                sd = {"c." + k: v for k, v in hashfill.fill_state_dict(cls(C, C).state_dict(), seed=seed, gain=0.6).items()}
                assert maxabs(R.dual_cell(sd, "c", x, s, deriv, gb), g[f"{key}_{tag}"]) <= TOL, (key, tag)


def test_unused_recurrent_modules():
    """the oracle's Dual_GRU / BiGRU / several-present-frames dual cells against the fixtures produced by
    the reference classes (tests/golden/unused_cells.npz)."""
    g = gold("unused_cells.npz")
    I = cases.unused_cell_inputs()
    fill = lambda shapes, key: hashfill.fill_state_dict(shapes, seed=cases.UNUSED_CELL_SEEDS[key], gain=0.6)
    from streamingflow_amd.layers.temporal import BiGRU, Dual_GRU
    from streamingflow_amd.layers.temporal_ode_bayes import DualGRUCell, DualGRUODECell
    with torch.no_grad():
        for mix in (True, False):
            sd = fill(Dual_GRU(8, 8, n_future=3, mixture=mix, gru_bias_init=0.3).state_dict(), "dual_gru")
            assert maxabs(R.dual_gru(sd, I["x1"], I["st1"], 3, mix, 0.3), g[f"dual_gru_mix{int(mix)}_p1"]) <= TOL
            assert maxabs(R.dual_gru(sd, I["x1"], I["st3"], 3, mix, 0.3), g[f"dual_gru_mix{int(mix)}_p3"]) <= TOL
        sd = fill(Dual_GRU(16, 8, n_future=2).state_dict(), "dual_gru_wide")
        assert maxabs(R.dual_gru(sd, I["x1w"], I["st1"], 2), g["dual_gru_wide"]) <= TOL
        sd = fill(BiGRU(8, gru_bias_init=-0.2).state_dict(), "bigru")
        assert maxabs(R.bigru(sd, I["seq"], -0.2), g["bigru"]) <= 2 * TOL
        sd = fill(DualGRUODECell(8, 8).state_dict(), "dual_ode")
        assert maxabs(R.dual_cell_frames(sd, I["x1b1"], I["st2b1"], True), g["dual_ode_p2"]) <= TOL
        sd = fill(DualGRUCell(8, 8).state_dict(), "dual_obs")
        assert maxabs(R.dual_cell_frames(sd, I["x1"], I["st3"], False), g["dual_obs_p3"]) <= TOL
