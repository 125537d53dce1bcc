"""Instance post-processing of StreamingFlow's evaluation on the MI355X: the functions of
streamingflow/utils/instance.py that ``evaluate.py`` calls, same names / arguments / returns.

Per-pixel work runs in libsfnative (``sf_instance_centers_fwd``: threshold + 3x3 NMS + ordered list,
``sf_group_pixels_fwd``: nearest-centre assignment with the foreground mask, ``sf_instance_sums_fwd``:
per-instance position sums for the temporal matching); the data-dependent control flow (id bookkeeping,
Hungarian assignment on a handful of centres via scipy) stays on the host as in the reference.
CUDA tensors only.
"""
from typing import Tuple

import numpy as np
import torch
from scipy.optimize import linear_sum_assignment

from . import _lib, runtime
from .runtime import ptr


def find_instance_centers(center_prediction: torch.Tensor, conf_threshold: float = 0.1, nms_kernel_size: float = 3):
    """instance.py:80-92.  center_prediction [1, H, W] -> [n, 2] int64 (row, col) in row-major order."""
    assert len(center_prediction.shape) == 3
    if nms_kernel_size != 3:
        raise NotImplementedError("only the 3x3 NMS the reference uses")
    runtime.require_cuda(center_prediction)
    c = runtime.f32c(center_prediction).view(center_prediction.shape[-2], center_prediction.shape[-1])
    H, W = c.shape
    L = _lib.lib()
    cap = H * W
    centers = torch.empty((cap, 2), dtype=torch.int32, device=c.device)
    n = torch.empty((), dtype=torch.int32, device=c.device)
    ws = runtime.workspace(L.sf_instance_centers_ws_bytes(H, W), c.device)
    _lib.check(L.sf_instance_centers_fwd(ptr(c), H, W, float(conf_threshold), ptr(centers), cap, ptr(n), ptr(ws), ws.numel() * 4,
                                         runtime.stream_ptr(c.device)), "instance_centers")
    return centers[: int(n.item())].long()


def group_pixels(centers: torch.Tensor, offset_predictions: torch.Tensor, foreground_mask: torch.Tensor = None) -> torch.Tensor:
    """instance.py:95-116 (+ the foreground product of :136 when a mask is given) -> [1, H, W] int64 ids from 1."""
    runtime.require_cuda(centers, offset_predictions)
    H, W = offset_predictions.shape[-2:]
    off = runtime.f32c(offset_predictions).view(2, H, W)
    fg = torch.ones((H, W), dtype=torch.uint8, device=off.device) if foreground_mask is None else \
        (foreground_mask.reshape(H, W) != 0).to(torch.uint8).contiguous()
    c32 = centers.to(torch.int32).contiguous()
    out = torch.empty((1, H, W), dtype=torch.int64, device=off.device)
    _lib.check(_lib.lib().sf_group_pixels_fwd(ptr(c32), c32.shape[0], ptr(off), ptr(fg), H, W, ptr(out), runtime.stream_ptr(off.device)),
               "group_pixels")
    return out


def update_instance_ids(instance_seg, old_ids, new_ids):
    """instance.py:143-160."""
    indices = torch.arange(int(old_ids.max()) + 1, device=instance_seg.device)
    indices[torch.as_tensor(old_ids, device=instance_seg.device).long()] = torch.as_tensor(new_ids, device=instance_seg.device).long()
    return indices[instance_seg].long()


def make_instance_seg_consecutive(instance_seg):
    """instance.py:163-168."""
    unique_ids = torch.unique(instance_seg)
    new_ids = torch.arange(len(unique_ids), device=instance_seg.device)
    return update_instance_ids(instance_seg, unique_ids, new_ids)


def get_instance_segmentation_and_centers(center_predictions, offset_predictions, foreground_mask, conf_threshold: float = 0.1,
                                          nms_kernel_size: float = 3, max_n_instance_centers: int = 100) -> Tuple[torch.Tensor, torch.Tensor]:
    """instance.py:119-140."""
    width, height = center_predictions.shape[-2:]
    center_predictions = center_predictions.view(1, width, height)
    offset_predictions = offset_predictions.view(2, width, height)
    foreground_mask = foreground_mask.view(1, width, height)
    centers = find_instance_centers(center_predictions, conf_threshold=conf_threshold, nms_kernel_size=nms_kernel_size)
    if not len(centers):
        return torch.zeros(center_predictions.shape, dtype=torch.int64, device=center_predictions.device), \
            torch.zeros((0, 2), device=centers.device)
    if len(centers) > max_n_instance_centers:
        centers = centers[:max_n_instance_centers].clone()
    instance_seg = group_pixels(centers, offset_predictions, foreground_mask)
    instance_seg = make_instance_seg_consecutive(instance_seg)
    return instance_seg.long(), centers


def _instance_means(inst, flow, max_id):
    """Mean (row + flow0, col + flow1) of every instance id 1..max_id -> ([max_id + 1, 2] float32, counts)."""
    H, W = inst.shape[-2:]
    dev = inst.device
    sums = torch.empty((max_id + 1, 2), dtype=torch.float64, device=dev)
    cnt = torch.empty((max_id + 1,), dtype=torch.int32, device=dev)
    i64 = inst.reshape(H, W).to(torch.int64).contiguous()
    fl = runtime.f32c(flow).view(2, H, W) if flow is not None else None
    _lib.check(_lib.lib().sf_instance_sums_fwd(ptr(i64), ptr(fl), H, W, int(max_id), ptr(sums), ptr(cnt), runtime.stream_ptr(dev)),
               "instance_sums")
    means = (sums / cnt.clamp(min=1).unsqueeze(1).double()).float()
    return means, cnt


def make_instance_id_temporally_consistent(pred_inst, future_flow, matching_threshold=3.0):
    """instance.py:171-263.  pred_inst [1, seq, h, w], future_flow [1, seq, 2, h, w] -> consistent ids [1, seq, h, w]."""
    assert pred_inst.shape[0] == 1, "Assumes batch size = 1"
    runtime.require_cuda(pred_inst, future_flow)
    consistent = [pred_inst[0, 0]]
    largest_instance_id = consistent[0].max().item()
    _, seq_len, h, w = pred_inst.shape
    for t in range(seq_len - 1):
        t_instance_ids = torch.unique(consistent[-1])[1:].cpu().numpy()
        if len(t_instance_ids) == 0:
            consistent.append(pred_inst[0, t + 1])
            continue
        means_t, _ = _instance_means(consistent[-1], future_flow[0, t], int(t_instance_ids.max()))
        warped_centers = means_t[torch.as_tensor(t_instance_ids, device=means_t.device).long()]
        n_instances = int(pred_inst[0, t + 1].max().item())
        if n_instances == 0:
            consistent.append(pred_inst[0, t + 1])
            continue
        centers, _ = _instance_means(pred_inst[0, t + 1], None, n_instances)
        centers = centers[1:]
        distances = torch.norm(centers.unsqueeze(0) - warped_centers.unsqueeze(1), dim=-1).cpu().numpy()
        ids_t, ids_t_one = linear_sum_assignment(distances)
        matching_distances = distances[ids_t, ids_t_one]
        ids_t += 1
        ids_t_one += 1
        id_mapping = dict(zip(np.arange(1, len(t_instance_ids) + 1), t_instance_ids))
        ids_t = np.vectorize(id_mapping.__getitem__, otypes=[np.int64])(ids_t)
        ids_t = ids_t[matching_distances < matching_threshold]
        ids_t_one = ids_t_one[matching_distances < matching_threshold]
        remaining_ids = set(torch.unique(pred_inst[0, t + 1]).cpu().numpy()).difference(set(ids_t_one))
        remaining_ids.remove(0)
        for remaining_id in list(remaining_ids):
            largest_instance_id += 1
            ids_t = np.append(ids_t, largest_instance_id)
            ids_t_one = np.append(ids_t_one, remaining_id)
        consistent.append(update_instance_ids(pred_inst[0, t + 1], old_ids=torch.as_tensor(ids_t_one), new_ids=torch.as_tensor(ids_t)))
    return torch.stack(consistent).unsqueeze(0)


def predict_instance_segmentation_and_trajectories(output, compute_matched_centers=False, make_consistent=True, vehicles_id=1):
    """instance.py:370-428."""
    preds = output["segmentation"].detach()
    preds = torch.argmax(preds, dim=2, keepdim=True)
    foreground_masks = preds.squeeze(2) == vehicles_id
    batch_size, seq_len = preds.shape[:2]
    pred_inst = []
    for b in range(batch_size):
        frames = []
        for t in range(seq_len):
            inst_t, _ = get_instance_segmentation_and_centers(output["instance_center"][b, t].detach(), output["instance_offset"][b, t].detach(),
                                                              foreground_masks[b, t].detach())
            frames.append(inst_t)
        pred_inst.append(torch.stack(frames, dim=0))
    pred_inst = torch.stack(pred_inst).squeeze(2)
    if make_consistent:
        if output["instance_flow"] is None:
            output["instance_flow"] = torch.zeros_like(output["instance_offset"])
        consistent = torch.cat([make_instance_id_temporally_consistent(pred_inst[b:b + 1], output["instance_flow"][b:b + 1].detach())
                                for b in range(batch_size)], dim=0)
    else:
        consistent = pred_inst
    if compute_matched_centers:
        assert batch_size == 1
        matched_centers = {}
        _, seq_len, h, w = consistent.shape
        for t in range(seq_len):
            max_id = int(consistent[0, t].max().item())
            if max_id == 0:
                continue
            means, cnt = _instance_means(consistent[0, t], None, max_id)
            means, cnt = means.cpu(), cnt.cpu()
            for instance_id in torch.unique(consistent[0, 0])[1:].cpu().numpy():
                if instance_id <= max_id and cnt[instance_id] > 0:
                    matched_centers[instance_id] = matched_centers.get(instance_id, []) + [means[instance_id]]
        for key, value in matched_centers.items():
            matched_centers[key] = torch.stack(value).numpy()[:, ::-1]
        return consistent, matched_centers
    return consistent
