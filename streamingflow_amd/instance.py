"""Instance post-processing of StreamingFlow's evaluation on the MI355X.

Drop-in for the functions of ``streamingflow/utils/instance.py`` that ``evaluate.py`` reaches (``find_instance_centers``
:80-92, ``group_pixels`` :95-116, ``get_instance_segmentation_and_centers`` :119-140, ``update_instance_ids`` :143-160,
``make_instance_seg_consecutive`` :163-168, ``make_instance_id_temporally_consistent`` :171-263,
``predict_instance_segmentation_and_trajectories`` :370-428): same names, arguments and results.

How it is computed here.
  * Centres, pixel grouping: one kernel each (``sf_instance_centers_fwd``, ``sf_group_pixels_fwd``).
  * Relabelling is a look-up table gathered on the device; "make ids consecutive" is the inverse index of a sorted
    unique.
  * Temporal consistency.  The ids a frame ends up with are a relabelling of its own raw instances, so everything the
    matching needs — each raw instance's pixel count, centre, and centre displaced by the predicted flow — is computed
    for ALL frames by one launch (``sf_instance_moments_fwd``: integer atomics, order-independent) and copied to the
    host once.  The frame-to-frame assignment (Hungarian method on a handful of centres, ``scipy``) then only produces
    one small table per frame, and the whole sequence is relabelled by a single gather.
CUDA tensors only.
"""
from typing import Tuple

import numpy as np
import torch
from scipy.optimize import linear_sum_assignment

from . import _lib, runtime
from .runtime import ptr

MOMENT_SCALE = 1.0 / 1048576.0      # sf_instance_moments_fwd: flow-warped sums are 2^-20 fixed point


def find_instance_centers(center_prediction: torch.Tensor, conf_threshold: float = 0.1, nms_kernel_size: float = 3):
    """[1, H, W] centre heat map -> [n, 2] int64 (row, col) of its thresholded 3x3 local maxima, row-major order."""
    if center_prediction.dim() != 3:
        raise AssertionError("center_prediction must be [1, H, W]")
    if nms_kernel_size != 3:
        raise NotImplementedError("only the 3x3 non-maximum suppression the reference uses is built")
    runtime.require_cuda(center_prediction)
    H, W = center_prediction.shape[-2:]
    heat = runtime.f32c(center_prediction).view(H, W)
    L = _lib.lib()
    found = torch.empty((H * W, 2), dtype=torch.int32, device=heat.device)
    count = torch.empty((), dtype=torch.int32, device=heat.device)
    ws = runtime.workspace(L.sf_instance_centers_ws_bytes(H, W), heat.device)
    _lib.check(L.sf_instance_centers_fwd(ptr(heat), H, W, float(conf_threshold), ptr(found), H * W, ptr(count), ptr(ws), ws.numel() * 4,
                                         runtime.stream_ptr(heat.device)), "instance_centers")
    return found[: int(count.item())].long()


def group_pixels(centers: torch.Tensor, offset_predictions: torch.Tensor, foreground_mask: torch.Tensor = None) -> torch.Tensor:
    """Every pixel votes for the centre nearest to (pixel + predicted offset): [1, H, W] int64 ids from 1 (first centre
    wins ties); pixels outside ``foreground_mask`` (when given) get 0."""
    runtime.require_cuda(centers, offset_predictions)
    H, W = offset_predictions.shape[-2:]
    votes = runtime.f32c(offset_predictions).view(2, H, W)
    if foreground_mask is None:
        inside = torch.ones((H, W), dtype=torch.uint8, device=votes.device)
    else:
        inside = (foreground_mask.reshape(H, W) != 0).to(torch.uint8).contiguous()
    table = centers.to(torch.int32).contiguous()
    ids = torch.empty((1, H, W), dtype=torch.int64, device=votes.device)
    _lib.check(_lib.lib().sf_group_pixels_fwd(ptr(table), table.shape[0], ptr(votes), ptr(inside), H, W, ptr(ids),
                                              runtime.stream_ptr(votes.device)), "group_pixels")
    return ids


def update_instance_ids(instance_seg, old_ids, new_ids):
    """Relabel: every id listed in ``old_ids`` becomes the matching entry of ``new_ids``, the others stay."""
    dev = instance_seg.device
    old = torch.as_tensor(old_ids, device=dev).long()
    table = torch.arange(int(old.max()) + 1, device=dev)
    table[old] = torch.as_tensor(new_ids, device=dev).long()
    return table[instance_seg].long()


def make_instance_seg_consecutive(instance_seg):
    """Ids become 0..n-1 in the order of their old values."""
    return torch.unique(instance_seg, return_inverse=True)[1].long()


def get_instance_segmentation_and_centers(center_predictions, offset_predictions, foreground_mask, conf_threshold: float = 0.1,
                                          nms_kernel_size: float = 3, max_n_instance_centers: int = 100) -> Tuple[torch.Tensor, torch.Tensor]:
    """One frame: ([1, H, W] int64 instance map with consecutive ids, [n, 2] centres)."""
    H, W = center_predictions.shape[-2:]
    heat = center_predictions.reshape(1, H, W)
    peaks = find_instance_centers(heat, conf_threshold=conf_threshold, nms_kernel_size=nms_kernel_size)
    if peaks.shape[0] == 0:
        return torch.zeros((1, H, W), dtype=torch.int64, device=heat.device), torch.zeros((0, 2), device=heat.device)
    peaks = peaks[:max_n_instance_centers].clone() if peaks.shape[0] > max_n_instance_centers else peaks
    ids = group_pixels(peaks, offset_predictions.reshape(2, H, W), foreground_mask.reshape(1, H, W))
    return make_instance_seg_consecutive(ids), peaks


def instance_moments(instance_seq, flow_seq=None):
    """instance_seq [F, H, W] int64, flow_seq [F, 2, H, W] or None -> numpy (counts [F, K], centres [F, K, 2] float32,
    flow-displaced centres [F, K, 2] float32 or None), K = largest id + 1; rows of absent ids are NaN."""
    runtime.require_cuda(instance_seq)
    F, H, W = instance_seq.shape
    ids = instance_seq.to(torch.int64).contiguous()
    top = int(ids.max().item())
    dev = ids.device
    pos = torch.empty((F, top + 1, 2), dtype=torch.int64, device=dev)
    cnt = torch.empty((F, top + 1), dtype=torch.int32, device=dev)
    fl = runtime.f32c(flow_seq).view(F, 2, H, W) if flow_seq is not None else None
    moved = torch.empty_like(pos) if fl is not None else None
    _lib.check(_lib.lib().sf_instance_moments_fwd(ptr(ids), ptr(fl), F, H, W, top, ptr(pos), ptr(moved), ptr(cnt), runtime.stream_ptr(dev)),
               "instance_moments")
    counts = cnt.cpu().numpy()
    with np.errstate(invalid="ignore", divide="ignore"):
        denom = counts[..., None].astype(np.float64)
        centres = (pos.cpu().numpy() / denom).astype(np.float32)
        displaced = (moved.cpu().numpy() * MOMENT_SCALE / denom).astype(np.float32) if moved is not None else None
    return counts, centres, displaced


def make_instance_id_temporally_consistent(pred_inst, future_flow, matching_threshold=3.0):
    """pred_inst [1, T, h, w] per-frame instance maps, future_flow [1, T, 2, h, w] -> [1, T, h, w] with ids that follow the
    instances through time: an instance of frame t+1 inherits the id of the frame-t instance whose flow-displaced centre
    it is assigned to (Hungarian method) when they are closer than ``matching_threshold``; otherwise it gets a new id."""
    assert pred_inst.shape[0] == 1, "Assumes batch size = 1"
    runtime.require_cuda(pred_inst, future_flow)
    frames = pred_inst[0]
    T = frames.shape[0]
    counts, centres, displaced = instance_moments(frames, future_flow[0])
    K = counts.shape[1]
    tables = np.tile(np.arange(K, dtype=np.int64), (T, 1))      # tables[t][raw id] = consistent id; frame 0 keeps its ids
    next_new = int(np.flatnonzero(counts[0]).max(initial=0))      # largest id of the first frame
    for t in range(T - 1):
        raw_prev = np.flatnonzero(counts[t][1:]) + 1
        raw_next = np.flatnonzero(counts[t + 1][1:]) + 1
        if raw_prev.size == 0 or raw_next.size == 0:
            continue                                              # nothing to carry over: frame t+1 keeps its raw ids
        order = np.argsort(tables[t][raw_prev], kind="stable")    # previous instances by ascending consistent id
        raw_prev = raw_prev[order]
        carried = tables[t][raw_prev]
        n_next = int(raw_next.max())                              # raw ids are consecutive 1..n_next
        gap = np.linalg.norm(centres[t + 1, 1:n_next + 1][None, :, :] - displaced[t, raw_prev][:, None, :], axis=-1)
        rows, cols = linear_sum_assignment(gap)
        close = gap[rows, cols] < matching_threshold
        table = tables[t + 1]
        table[cols[close] + 1] = carried[rows[close]]
        unmatched = np.setdiff1d(raw_next, cols[close] + 1)       # ascending
        table[unmatched] = next_new + 1 + np.arange(unmatched.size)
        next_new += int(unmatched.size)
    lut = torch.from_numpy(tables).to(frames.device)
    return torch.gather(lut, 1, frames.reshape(T, -1).long()).view_as(frames).unsqueeze(0)


def predict_instance_segmentation_and_trajectories(output, compute_matched_centers=False, make_consistent=True, vehicles_id=1):
    """Decoder output dict (``segmentation`` [b, T, classes, H, W], ``instance_center`` [b, T, 1, H, W], ``instance_offset`` and
    ``instance_flow`` [b, T, 2, H, W]) -> [b, T, H, W] int64 instance ids (+ {id: [n, 2] (x, y) centre track} of the
    instances of the first frame when ``compute_matched_centers``, batch size 1)."""
    labels = output["segmentation"].detach().argmax(dim=2)
    vehicles = labels == vehicles_id
    B, T = labels.shape[:2]
    centre_maps, offsets = output["instance_center"].detach(), output["instance_offset"].detach()
    per_frame = torch.stack([torch.stack([get_instance_segmentation_and_centers(centre_maps[b, t], offsets[b, t], vehicles[b, t])[0][0]
                                          for t in range(T)]) for b in range(B)])
    if make_consistent:
        if output["instance_flow"] is None:
            output["instance_flow"] = torch.zeros_like(output["instance_offset"])
        flow = output["instance_flow"].detach()
        tracked = torch.cat([make_instance_id_temporally_consistent(per_frame[b:b + 1], flow[b:b + 1]) for b in range(B)])
    else:
        tracked = per_frame
    if not compute_matched_centers:
        return tracked
    assert B == 1
    counts, centres, _ = instance_moments(tracked[0])
    tracks = {}
    for ident in (np.flatnonzero(counts[0][1:]) + 1).tolist():      # the instances of the first frame, wherever they reappear
        seen = counts[:, ident] > 0
        tracks[ident] = centres[seen, ident][:, ::-1]               # (row, col) -> (x, y)
    return tracked, tracks
