"""Label preparation of StreamingFlow's evaluation on the MI355X.

Drop-in for ``warp_features`` / ``cumulative_warp_features`` / ``cumulative_warp_features_reverse``
(``streamingflow/utils/geometry.py:196-297``), their helpers ``mat2pose_vec`` / ``invert_pose_matrix`` (:97-121, :175-193)
and ``prepare_future_labels`` (``streamingflow/trainer.py:283-394``; a function of (batch, cfg, receptive_field,
spatial_extent, encoder_downsample) here — the reference method lives on a pytorch_lightning module).

How it is computed here.  Warping a BEV map by an ego-motion is a planar rigid transform: yaw about z plus an (x, y)
shift.  For a sequence the ego-motions are chained as 4x4 matrices first (all samples at once, a handful of tiny
products), each chained matrix is reduced to its (yaw, shift) and turned into one 2x3 sampling matrix, and then ALL
frames of the sequence are resampled by a single launch of ``sf_warp_affine_fwd`` (affine grid + nearest / bilinear
sampling, zeros outside, ``align_corners=False``).  CUDA tensors only.
"""
import torch

from . import _lib, runtime
from .models.lift_splat import pose_vec2mat
from .runtime import ptr


def mat2pose_vec(matrix):
    """[..., 4, 4] rigid transform -> [..., 6] (x, y, z, rot_x, rot_y, rot_z), Euler angles of R = Rx Ry Rz."""
    r = matrix[..., :3, :3]
    rot_x = torch.atan2(-r[..., 1, 2], r[..., 2, 2])
    rot_y = torch.atan2(r[..., 0, 2], torch.sqrt(r[..., 1, 2] ** 2 + r[..., 2, 2] ** 2))
    rot_z = torch.atan2(-r[..., 0, 1], r[..., 0, 0])
    return torch.cat((matrix[..., :3, 3], torch.stack((rot_x, rot_y, rot_z), dim=-1)), dim=-1)


def invert_pose_matrix(x):
    """[n, 4, 4] rigid transforms -> their inverses ([R | t]^-1 = [R^T | -R^T t])."""
    assert x.dim() == 3 and x.shape[1:] == (4, 4), "Only works for batch of pose matrices."
    inv = torch.zeros_like(x)
    rt = x[:, :3, :3].transpose(1, 2)
    inv[:, :3, :3] = rt
    inv[:, :3, 3:] = -torch.bmm(rt, x[:, :3, 3:])
    inv[:, 3, 3] = 1.0
    return inv


def _sampling_matrices(pose, spatial_extent):
    """[n, 6] pose vectors -> [n, 6] row-major 2x3 matrices for F.affine_grid-style sampling: rotation by the yaw, the
    metric shift normalised by the half extents of the grid (x is mirrored: image rows grow towards -x)."""
    yaw = pose[:, 5]
    c, s = torch.cos(yaw), torch.sin(yaw)
    shift_rows = -(pose[:, 0] / spatial_extent[0])
    shift_cols = pose[:, 1] / spatial_extent[1]
    return torch.stack([c, -s, shift_cols, s, c, shift_rows], dim=-1).float().contiguous()


def _resample(maps, theta, mode):
    """maps [n, c, h, w], theta [n, 6] -> resampled maps, one launch for all n."""
    if mode not in ("nearest", "bilinear"):
        raise ValueError(mode)
    runtime.require_cuda(maps, theta)
    n, c, h, w = maps.shape
    src = runtime.f32c(maps)
    dst = torch.empty_like(src)
    _lib.check(_lib.lib().sf_warp_affine_fwd(ptr(src), ptr(theta), n, c, h, w, int(mode == "bilinear"), ptr(dst), runtime.stream_ptr(maps.device)),
               "warp_affine")
    return dst.to(maps.dtype)


def warp_features(x, flow, mode="nearest", spatial_extent=None):
    """x [b, c, h, w] BEV maps, flow [b, 6] ego-motion pose vectors -> the maps seen from the moved ego frame."""
    if flow is None:
        return x
    return _resample(x, _sampling_matrices(flow, spatial_extent), mode)


def _warp_sequence(x, chained, keep, mode, spatial_extent):
    """x [b, T, c, h, w]; chained[t]: [b, 4, 4] transform of frame t (None for the frames listed in ``keep``, which are
    copied): every other frame is resampled, all of them in one launch."""
    b, T = x.shape[:2]
    moved = [t for t in range(T) if t not in keep]
    out = x.clone()
    if moved:
        poses = mat2pose_vec(torch.stack([chained[t] for t in moved], dim=1)).reshape(b * len(moved), 6)
        maps = x[:, moved].reshape((b * len(moved),) + tuple(x.shape[2:]))
        out[:, moved] = _resample(maps, _sampling_matrices(poses, spatial_extent), mode).view((b, len(moved)) + tuple(x.shape[2:]))
    return out


def cumulative_warp_features(x, flow, mode="nearest", spatial_extent=None):
    """Past frames into the LAST frame's ego frame: frame t is moved by flow[t] ... flow[T-2] chained (flow[t] takes frame
    t to frame t+1)."""
    T = x.shape[1]
    if T == 1:
        return x
    step = pose_vec2mat(flow)
    chained = {T - 2: step[:, T - 2]}
    for t in range(T - 3, -1, -1):
        chained[t] = step[:, t] @ chained[t + 1]
    return _warp_sequence(x, chained, {T - 1}, mode, spatial_extent)


def cumulative_warp_features_reverse(x, flow, mode="nearest", spatial_extent=None):
    """Future frames into the FIRST frame's ego frame: frame i is moved back by the inverses of flow[0] ... flow[i-1]."""
    T = x.shape[1]
    step = pose_vec2mat(flow)
    chained = {}
    for i in range(1, T):
        back = invert_pose_matrix(step[:, i - 1])
        chained[i] = back if i == 1 else chained[i - 1] @ back
    return _warp_sequence(x, chained, {0}, mode, spatial_extent)


def prepare_future_labels(batch, cfg, receptive_field, spatial_extent, encoder_downsample=8, is_lyft=False):
    """Every label sequence of a batch in the PRESENT frame (index receptive_field - 1): the past is warped forward, the
    future backward, and the two halves are joined without repeating the present."""
    rf = receptive_field
    ego = batch["future_egomotion"]
    labels = {}
    if not is_lyft and "gt_trajectory" in batch:
        labels["gt_trajectory"] = batch["gt_trajectory"]
    if cfg.LIFT.GT_DEPTH and "depths" in batch:
        lo, hi = cfg.LIFT.D_BOUND[0], cfg.LIFT.D_BOUND[1]
        bins = batch["depths"][:, :rf, :, ::encoder_downsample, ::encoder_downsample].clamp(lo, hi - 1) - lo
        labels["depths"] = bins.long().contiguous()

    def to_present(seq, categorical, channel_less=False):
        maps = seq.float() if categorical else seq
        if channel_less:
            maps = maps.unsqueeze(2)
        past = cumulative_warp_features(maps[:, :rf], ego[:, :rf], mode="nearest", spatial_extent=spatial_extent)[:, :-1]
        future = cumulative_warp_features_reverse(maps[:, rf - 1:], ego[:, rf - 1:], mode="nearest", spatial_extent=spatial_extent)
        joined = torch.cat([past, future], dim=1)
        if categorical:
            joined = joined.long()
        return joined[:, :, 0].contiguous() if channel_less else joined.contiguous()

    labels["segmentation"] = to_present(batch["segmentation"], True)
    if cfg.SEMANTIC_SEG.PEDESTRIAN.ENABLED:
        labels["pedestrian"] = to_present(batch["pedestrian"], True)
    if cfg.INSTANCE_SEG.ENABLED:
        labels["instance"] = to_present(batch["instance"], True, channel_less=True)
        labels["centerness"] = to_present(batch["centerness"], False)
        labels["offset"] = to_present(batch["offset"], False)
    if cfg.INSTANCE_FLOW.ENABLED:
        labels["flow"] = to_present(batch["flow"], False)
    return labels
