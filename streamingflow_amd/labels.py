"""Label preparation of StreamingFlow's evaluation on the MI355X: ``warp_features``,
``cumulative_warp_features``, ``cumulative_warp_features_reverse`` (streamingflow/utils/geometry.py:196-297) and
``prepare_future_labels`` (streamingflow/trainer.py:283-394, here a function of (batch, cfg, receptive_field,
spatial_extent, encoder_downsample) — the reference method needs pytorch_lightning).

The warp itself (affine grid + nearest / bilinear sampling) is one kernel (``sf_warp_affine_fwd``); composing the
6-DoF poses is a handful of 4x4 products on [b, t] matrices (torch, as in the reference).  CUDA tensors only.
"""
import torch

from . import _lib, runtime
from .models.lift_splat import pose_vec2mat
from .runtime import ptr


def mat2pose_vec(matrix):
    """utils/geometry.py:97-121."""
    rotx = torch.atan2(-matrix[..., 1, 2], matrix[..., 2, 2])
    cosy = torch.sqrt(matrix[..., 1, 2] ** 2 + matrix[..., 2, 2] ** 2)
    roty = torch.atan2(matrix[..., 0, 2], cosy)
    rotz = torch.atan2(-matrix[..., 0, 1], matrix[..., 0, 0])
    return torch.cat((matrix[..., :3, 3], torch.stack((rotx, roty, rotz), dim=-1)), dim=-1)


def invert_pose_matrix(x):
    """utils/geometry.py:175-193."""
    assert len(x.shape) == 3 and x.shape[1:] == (4, 4), "Only works for batch of pose matrices."
    rt = torch.transpose(x[:, :3, :3], 1, 2)
    inv = torch.cat([rt, -torch.bmm(rt, x[:, :3, 3:])], dim=-1)
    inv = torch.nn.functional.pad(inv, [0, 0, 0, 1], value=0)
    inv[..., 3, 3] = 1.0
    return inv


def warp_features(x, flow, mode="nearest", spatial_extent=None):
    """utils/geometry.py:196-236.  x [b, c, h, w], flow [b, 6] -> warped [b, c, h, w]."""
    if flow is None:
        return x
    if mode not in ("nearest", "bilinear"):
        raise ValueError(mode)
    runtime.require_cuda(x, flow)
    b, c, h, w = x.shape
    angle = flow[:, 5].clone()
    translation = flow[:, :2].clone()
    translation[:, 0] /= spatial_extent[0]
    translation[:, 1] /= spatial_extent[1]
    translation[:, 0] *= -1
    cos_theta, sin_theta = torch.cos(angle), torch.sin(angle)
    theta = torch.stack([cos_theta, -sin_theta, translation[:, 1], sin_theta, cos_theta, translation[:, 0]], dim=-1).float().contiguous()
    xf = runtime.f32c(x)
    out = torch.empty_like(xf)
    _lib.check(_lib.lib().sf_warp_affine_fwd(ptr(xf), ptr(theta), b, c, h, w, int(mode == "bilinear"), ptr(out), runtime.stream_ptr(x.device)),
               "warp_affine")
    return out.to(x.dtype)


def cumulative_warp_features(x, flow, mode="nearest", spatial_extent=None):
    """utils/geometry.py:239-267."""
    sequence_length = x.shape[1]
    if sequence_length == 1:
        return x
    flow = pose_vec2mat(flow)
    out = [x[:, -1]]
    cum_flow = flow[:, -2]
    for t in reversed(range(sequence_length - 1)):
        out.append(warp_features(x[:, t], mat2pose_vec(cum_flow), mode=mode, spatial_extent=spatial_extent))
        cum_flow = flow[:, t - 1] @ cum_flow
    return torch.stack(out[::-1], 1)


def cumulative_warp_features_reverse(x, flow, mode="nearest", spatial_extent=None):
    """utils/geometry.py:270-294."""
    flow = pose_vec2mat(flow)
    out = [x[:, 0]]
    for i in range(1, x.shape[1]):
        cum_flow = invert_pose_matrix(flow[:, 0]) if i == 1 else cum_flow @ invert_pose_matrix(flow[:, i - 1])
        out.append(warp_features(x[:, i], mat2pose_vec(cum_flow), mode, spatial_extent=spatial_extent))
    return torch.stack(out, 1)


def prepare_future_labels(batch, cfg, receptive_field, spatial_extent, encoder_downsample=8, is_lyft=False):
    """trainer.py:283-394: warp every label sequence into the present frame."""
    labels = {}
    seg = batch["segmentation"]
    ego = batch["future_egomotion"]
    rf = receptive_field
    if not is_lyft and "gt_trajectory" in batch:
        labels["gt_trajectory"] = batch["gt_trajectory"]
    if cfg.LIFT.GT_DEPTH and "depths" in batch:
        d = batch["depths"][:, :rf, :, ::encoder_downsample, ::encoder_downsample]
        d = torch.clamp(d, cfg.LIFT.D_BOUND[0], cfg.LIFT.D_BOUND[1] - 1) - cfg.LIFT.D_BOUND[0]
        labels["depths"] = d.long().contiguous()

    def both(x, to_long, squeeze=False):
        xin = x.float().unsqueeze(2) if squeeze else x.float() if to_long else x
        past = cumulative_warp_features(xin[:, :rf], ego[:, :rf], mode="nearest", spatial_extent=spatial_extent)
        fut = cumulative_warp_features_reverse(xin[:, (rf - 1):], ego[:, (rf - 1):], mode="nearest", spatial_extent=spatial_extent)
        if to_long:
            past, fut = past.long(), fut.long()
        past, fut = past.contiguous()[:, :-1], fut.contiguous()
        if squeeze:
            past, fut = past[:, :, 0], fut[:, :, 0]
        return torch.cat([past, fut], dim=1)
    labels["segmentation"] = both(seg, True)
    if cfg.SEMANTIC_SEG.PEDESTRIAN.ENABLED:
        labels["pedestrian"] = both(batch["pedestrian"], True)
    if cfg.INSTANCE_SEG.ENABLED:
        labels["instance"] = both(batch["instance"], True, squeeze=True)
        labels["centerness"] = both(batch["centerness"], False)
        labels["offset"] = both(batch["offset"], False)
    if cfg.INSTANCE_FLOW.ENABLED:
        labels["flow"] = both(batch["flow"], False)
    return labels
