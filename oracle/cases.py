"""TEST INFRASTRUCTURE — shared definitions of the parity cases.

Both ``oracle/gen_golden.py`` (which runs the real reference in the build container) and the tests
(which run the oracle / the HIP path) regenerate inputs and weights from these definitions with
``oracle.hashfill``; only expected outputs are stored under ``tests/golden``.
"""
import torch

from . import hashfill

WEIGHT_SEED = 1
# fan-in-scaled uniform weights times a per-subtree gain, tuned so that every tensor on the path
# stays O(1) over 46+ chained steps while the outputs remain sensitive to eps / IMPUTE / dt.
WEIGHT_GAINS = (("srvp_encoder", 1.3), ("srvp_decoder", 1.1), ("gru_ode", 1.5))
WEIGHT_GAIN_DEFAULT = 1.0
EPS_SEED = 3

# name -> (camera ts, lidar ts, target ts, delta_t)
TIMESETS = {
    "shipped":      ([-1, -.5, 0], [-.8, -.6, -.4, -.2, 0], [-1, -.5, 0, .5, 1, 1.5, 2], 0.05),
    "config1":      ([0.0], [], [0.05, 0.10, 0.15, 0.20], 0.05),
    "future16":     ([-1, -.5, 0], [-.8, -.6, -.4, -.2, 0], [-1, -.5, 0] + [0.5 * k for k in range(1, 17)], 0.05),
    "stream40":     ([-1, -.5, 0], [-.8, -.6, -.4, -.2, 0], [-1, -.5, 0] + [0.05 * k for k in range(1, 41)], 0.05),
    "ties":         ([-.5, 0], [-.5, 0], [0, .5], 0.05),
    "tiny_gaps":    ([-.03, 0], [-.02], [0.02, 0.04, 0.1], 0.05),
    "past_only":    ([-1, -.5, 0], [-.8, -.2], [-1, -.5], 0.05),
    "accum_edge":   ([0.0], [], [0.15, 0.3, 0.35], 0.05),
    "accum_edge2":  ([-.3, 0.0], [-.15], [0.1, 0.7, 0.75, 1.05], 0.05),
    "unsorted_T":   ([-.5, 0], [-.2], [1.0, 0.5, 0.25], 0.05),
    "camera_only":  ([-1, -.5, 0], [], [0, .5, 1], 0.05),
    "lidar_only":   ([], [-.8, -.6, -.4, -.2, 0], [0, .5, 1], 0.05),
    "dt_010":       ([-1, -.5, 0], [-.8, -.6, -.4, -.2, 0], [0, .25, .5, .6], 0.1),
    "dt_025":       ([-1, -.5, 0], [-.75, -.25], [0, .25, .5, 1.1], 0.25),
    "target_now":   ([-.5, 0], [0], [0.0], 0.05),
    "irregular":    ([-.97, -.52, -.01], [-.93, -.71, -.33, -.07], [0.13, 0.49, 0.51, 1.27], 0.05),
    "late_lidar":   ([-1, -.5], [-.45, -.4, -.35, 0], [0.05, 0.1], 0.05),
    "datastream15": ([-1, -.5, 0], [-.9, -.75, -.6, -.45, -.3, -.15, 0], [.5, 1, 1.5, 2], 0.05),
    "datastream50": ([-1, -.5, 0], [-1, -.5, 0], [.5, 1, 1.5, 2], 0.05),
    "interval06":   ([-1, -.5, 0], [-.8, -.6, -.4, -.2, 0], [0.6, 1.2, 1.8], 0.05),
    "half_dt":      ([0.0], [], [0.025, 0.05, 0.075], 0.05),
    "single_far":   ([0.0], [], [8.0], 0.05),
}


def timeset(name):
    cam, lid, tgt, dt = TIMESETS[name]
    f = lambda v: torch.tensor([v], dtype=torch.float64).reshape(1, len(v))
    return f(cam), f(lid), f(tgt), dt


def bev_inputs(C, H, W, n_cam, n_lid, seed=0):
    """Synthetic encoder features ~N(0,1): camera (1,n_cam,C,H,W), lidar (1,n_lid,C,H,W)."""
    cam = hashfill.normal("cam", (1, n_cam, C, H, W), seed)
    lid = hashfill.normal("lid", (1, n_lid, C, H, W), seed)
    return cam, lid


def present_input(cam, lid):
    """future_prediction_input: the reference passes the present-frame state (1,1,C,H,W); its
    value is numerically unused (SURVEY.md §3.2 note), only its shape matters."""
    src = cam if cam.shape[1] else lid
    return src[:, -1:].clone()


def fpode_state_dict(shapes_sd, seed=WEIGHT_SEED):
    """Hashed weights for a FuturePredictionODE-shaped state_dict (keys with or without the
    checkpoint prefix ``model.future_prediction_ode.``)."""
    out = {}
    for k, v in shapes_sd.items():
        g = WEIGHT_GAIN_DEFAULT
        for frag, gg in WEIGHT_GAINS:
            if frag in k:
                g = gg
                break
        out.update(hashfill.fill_state_dict({k: v}, seed=seed, gain=g))
    return out


# Full-forward golden cases: name -> (C, H, W, timeset, solver, impute, variable, eps zero?)
FPODE_CASES = {
    "c8_16_shipped":        (8, 16, 16, "shipped", "euler", True, True, False),
    "c8_16_shipped_eps0":   (8, 16, 16, "shipped", "euler", True, True, True),
    "c8_16_fixed":          (8, 16, 16, "shipped", "euler", True, False, False),
    "c8_16_noimpute":       (8, 16, 16, "shipped", "euler", False, True, False),
    "c8_16_midpoint":       (8, 16, 16, "shipped", "midpoint", True, True, False),
    "c8_16_midpoint_fixed": (8, 16, 16, "dt_010", "midpoint", True, False, False),
    "c8_48_shipped":        (8, 48, 48, "shipped", "euler", True, True, False),
    "c8_48_config1":        (8, 48, 48, "config1", "euler", True, False, False),
    "c16_24_irregular":     (16, 24, 24, "irregular", "euler", True, True, False),
    "c8_16_lidar_only":     (8, 16, 16, "lidar_only", "euler", True, True, False),
}


# BEVerse-named secondary classes: tag -> (in_channels, latent_dim, h, w, n_future)
BEVERSE_CASES = {
    "config1_c32": (32, 16, 50, 50, 4),     # BASELINE config 1: FuturePrediction(32, 16, 3, 3) at 50x50
    "odd_c16": (16, 8, 13, 21, 2),          # odd sizes: zero-pad + ceil pooling of the skip path
}
