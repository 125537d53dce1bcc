"""Synthetic workloads of the hot path: timestamp sets, hashed inputs and hashed weights.

Neutral ground between the product's benches (``bench.py``, ``tools/``) and the checker (``oracle/``, ``tests/``):
both regenerate identical inputs and weights from ``(name, shape, seed)`` with ``workloads.hashfill``; nothing here
computes any part of the path.  ``oracle.cases`` re-exports these names next to its parity-case tables.
"""
from types import SimpleNamespace

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




def make_cfg(C, impute=True, solver="euler", variable=True, filter_size=None, skipco=False):
    """The 7 config keys the hot path reads (SURVEY.md §5 "Config / flags")."""
    return SimpleNamespace(MODEL=SimpleNamespace(
        IMPUTE=impute, SOLVER=solver,
        FUTURE_PRED=SimpleNamespace(USE_VARIABLE_ODE_STEP=variable),
        SMALL_ENCODER=SimpleNamespace(FILTER_SIZE=filter_size or C, SKIPCO=skipco),
        ENCODER=SimpleNamespace(OUT_CHANNELS=C)))


# ---- next rows: shipped configurations as data ------------------------------------------------------------
# LiDAR hard voxelisation (streamingflow.py:111): voxel size, range, <= 10 points per voxel, <= 160000 voxels (eval)
VOXEL_SHIPPED = ((0.0625, 0.0625, 0.2), (-50.0, -50.0, -5.0, 50.0, 50.0, 3.0), 10, 160000)
# SparseEncoder as streamingflow.py:111 builds it (block_type 'basicblock')
SPARSE_SHIPPED = dict(in_channels=5, sparse_shape=[1600, 1600, 41], output_channels=128, base_channels=16,
                      encoder_channels=[[16, 16, 32], [32, 32, 64], [64, 64, 128], [128, 128]],
                      encoder_paddings=[[0, 0, 1], [0, 0, 1], [0, 0, [1, 1, 0]], [0, 0]])
# Decoder: in_channels, n_classes, n_present, n_hdmap, task gates (the shipped heads)
DECODER_SHIPPED = (64, 2, 3, 2, dict(perceive_hdmap=False, predict_pedestrian=False, predict_instance=True,
                                     predict_future_flow=True, planning=False))


def decoder_state_dict(shapes_sd, seed=61):
    """Hashed weights; BatchNorm statistics randomised so that the folding is exercised (bn2.weight is NOT
    left at its zero init: that would silence the residual branches)."""
    return hashfill.fill_state_dict(shapes_sd, seed=seed, gain=0.9)
