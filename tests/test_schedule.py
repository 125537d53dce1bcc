"""CPU: the host scheduler (streamingflow_amd.schedule) against the schedules captured from the
reference's own control flow (tests/golden/schedules.json, temporal_ode_bayes.py:508-620) and
against the oracle's trace."""
import json
import os

import numpy as np
import pytest
import torch

from util import GOLD, cases, hashfill, build_pair
from streamingflow_amd import schedule as S
from streamingflow_amd._lib import OP_JUMP, SF_COEF_STRIDE


def _golden():
    with open(os.path.join(GOLD, "schedules.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("name", sorted(_golden()))
def test_schedule_matches_reference(name):
    e = _golden()[name]
    times, order = S.merge_observations(e["camera_ts"], e["lidar_ts"])
    sc = S.build_schedule(times, e["target_ts"], e["delta_t"], e["variable"])
    ops = [["jump", None] if k == OP_JUMP else ["step", sc.dts[a]] for k, a in sc.ops]
    assert ops == e["ops"]                      # bit-exact float64 dt values
    assert sc.sel_nops == e["select_nops"]
    assert [["cam" if s == 0 else "lidar", i] for s, i in order] == e["obs_order"]
    assert sc.n_steps == e["n_steps"] and sc.n_jumps == e["n_jumps"]


def test_known_answer_counts():
    g = _golden()
    assert g["shipped/variable"]["n_steps"] == 10 and g["shipped/variable"]["n_jumps"] == 8
    assert g["shipped/fixed"]["n_steps"] == 60
    assert g["stream40/variable"]["n_steps"] == 46


def test_draw_counts_and_coefficients():
    times, _ = S.merge_observations([-1, -.5, 0], [-.8, -.6, -.4, -.2, 0])
    for solver, per in (("euler", 1), ("midpoint", 2), ("rk4", 4)):
        sc = S.build_schedule(times, [-1, -.5, 0, .5, 1, 1.5, 2], 0.05, True, solver)
        assert sc.n_draws == sc.n_jumps + per * sc.n_steps
        c = sc.coef_array()
        assert c.shape == (sc.n_steps, SF_COEF_STRIDE) and c.dtype == np.float32
        assert c[0, 0] == np.float32(sc.dts[0]) and c[0, 1] == np.float32(sc.dts[0] / 2)
        assert c[0, 2] == np.float32(sc.dts[0] / 6) and c[0, 11] == 0


def test_empty_observations_raise():
    with pytest.raises(ValueError):
        S.build_schedule([], [0.5], 0.05, True)


def test_schedule_matches_oracle_trace():
    from oracle import ref_torch as R
    C, H = 8, 16
    for ts in ("irregular", "tiny_gaps", "unsorted_T"):
        for variable in (True, False):
            cts, lts, tts, dt = cases.timeset(ts)
            _, sd = build_pair(C, device="cpu")
            cam, lid = cases.bev_inputs(C, H, H, cts.shape[1], lts.shape[1])
            times, obs = R.merge_observations(cam, lid, cts, lts, 0)
            trace = []
            with torch.no_grad():
                R.nnfo_forward(sd, "gru_ode", times, cam[:, -1:], obs, dt, tts[0], "euler", True, variable,
                               hashfill.HashedNoise(0, zero=True), trace=trace)
            sc = S.build_schedule(times.tolist(), tts[0].tolist(), dt, variable)
            want = [("jump", None) if k == OP_JUMP else ("step", sc.dts[a]) for k, a in sc.ops]
            got = [(k, None if k == "jump" else v) for k, v in trace if k != "select"]
            assert got == want
