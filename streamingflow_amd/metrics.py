"""IoU and panoptic-quality accumulators for StreamingFlow's evaluation on the MI355X.

Drop-in for ``streamingflow/metrics.py`` (``IntersectionOverUnion`` :15-71, ``PanopticMetric`` :74-261): same
constructor arguments, ``update`` / ``compute`` / ``__call__`` / ``reset`` and state names, no ``pytorch_lightning``.

How it is computed here.  Every statistic of both metrics is a function of joint label histograms:
  * IoU: tp / fp / fn / support of class c are the diagonal, column and row sums of the (target x prediction) histogram.
  * PQ: a frame's segments are the instance ids themselves (id 0 = background "stuff", ids > 0 = vehicle "things");
    the histogram of (ground-truth id, predicted id) gives every intersection, every segment area (its margins) and
    hence every IoU.  A pair is a match when IoU > 0.5; ground-truth / predicted things that are in no match are the
    false negatives / false positives; a matched vehicle whose ground-truth id was matched to a different predicted id
    in an earlier frame is an id switch and counts as one false negative plus one false positive (:190-197).
The histograms of a whole [batch, time] block come from ONE kernel launch (``sf_confusion_frames_fwd``, integer
atomics: exact) and one device -> host copy; the bookkeeping on the resulting few-dozen-entry matrices is numpy.
"""
from typing import Optional

import numpy as np
import torch
import torch.nn as nn

from . import _lib, runtime
from .runtime import ptr

THING_CLASS = 1     # PanopticMetric derives the semantic map as (instance > 0): every instance is of class 1


def _as_labels(t):
    return t.reshape(-1).to(torch.int64).contiguous()


def confusion(pred, target, K):
    """([K, K] int64 with entry [t, p] = number of positions labelled t in ``target`` and p in ``pred``, device flag
    that is non-zero when a label was outside [0, K))."""
    runtime.require_cuda(pred, target)
    a, b = _as_labels(pred), _as_labels(target)
    if a.numel() != b.numel():
        raise ValueError("prediction and target must have the same number of elements")
    hist = torch.empty((K, K), dtype=torch.int64, device=a.device)
    flag = torch.empty((1,), dtype=torch.int32, device=a.device)
    _lib.check(_lib.lib().sf_confusion_fwd(ptr(a), ptr(b), a.numel(), int(K), ptr(hist), ptr(flag), runtime.stream_ptr(a.device)), "confusion")
    return hist, flag


def frame_confusions(pred, target, n_frames, K):
    """Per-frame joint histograms [n_frames, K, K] of two label tensors whose leading dims flatten to n_frames."""
    runtime.require_cuda(pred, target)
    a, b = _as_labels(pred), _as_labels(target)
    if a.numel() != b.numel() or a.numel() % n_frames:
        raise ValueError("prediction and target must hold n_frames equally sized maps")
    hist = torch.empty((n_frames, K, K), dtype=torch.int64, device=a.device)
    flag = torch.empty((1,), dtype=torch.int32, device=a.device)
    _lib.check(_lib.lib().sf_confusion_frames_fwd(ptr(a), ptr(b), a.numel() // n_frames, int(n_frames), int(K), ptr(hist), ptr(flag),
                                                  runtime.stream_ptr(a.device)), "confusion_frames")
    return hist, flag


class _Accumulator(nn.Module):
    """Named float32 counters that follow the data to its device, reset to zero and sum over ranks."""

    def __init__(self, names, size):
        super().__init__()
        self._counter_names = tuple(names)
        for name in self._counter_names:
            self.register_buffer(name, torch.zeros(size), persistent=False)

    def _accumulate(self, name, increment):
        state = getattr(self, name)
        if state.device != increment.device:        # first update on another device: the state moves, not the data
            for n in self._counter_names:
                setattr(self, n, getattr(self, n).to(increment.device))
            state = getattr(self, name)
        state += increment.to(state.dtype)

    def reset(self):
        for name in self._counter_names:
            getattr(self, name).zero_()

    def forward(self, *args, **kwargs):
        self.update(*args, **kwargs)

    def sync(self):
        """Sum the counters over the ranks of the default process group (the reference's ``dist_reduce_fx='sum'``)."""
        from . import dist
        for name in self._counter_names:
            dist.reduce_counters(getattr(self, name))


class IntersectionOverUnion(_Accumulator):
    def __init__(self, n_classes: int, ignore_index: Optional[int] = None, absent_score: float = 0.0, reduction: str = "none",
                 compute_on_step: bool = False):
        super().__init__(("true_positive", "false_positive", "false_negative", "support"), n_classes)
        if reduction not in ("none", "sum", "elementwise_mean"):
            raise ValueError("Reduction parameter unknown.")
        self.n_classes, self.ignore_index, self.absent_score, self.reduction = n_classes, ignore_index, absent_score, reduction
        self._out_of_range = None       # device flag, OR-ed over every update since the last reset

    def update(self, prediction: torch.Tensor, target: torch.Tensor):
        if prediction.ndim == target.ndim + 1:      # class scores: the predicted label is the arg max over dim 1
            prediction = prediction.argmax(dim=1)
        n = self.n_classes
        hist, flag = confusion(prediction, target, n)          # any label outside [0, n) — n itself included — trips the flag (compute() raises)
        self._out_of_range = flag if self._out_of_range is None else torch.maximum(self._out_of_range, flag)
        hist = hist.to(torch.float32)                           # rows = target classes 0..n-1
        hit = hist.diagonal()
        self._accumulate("true_positive", hit)
        self._accumulate("false_positive", hist.sum(0) - hit)
        self._accumulate("false_negative", hist.sum(1) - hit)
        self._accumulate("support", hist.sum(1))

    def reset(self):
        super().reset()
        self._out_of_range = None

    def compute(self):
        if self._out_of_range is not None and int(self._out_of_range.item()):
            raise RuntimeError("IntersectionOverUnion: a label outside [0, n_classes) was seen")
        tp, fp, fn = self.true_positive, self.false_positive, self.false_negative
        seen = (self.support + tp + fp) > 0
        scores = torch.where(seen, tp / (tp + fp + fn).clamp(min=1), torch.full_like(tp, float(self.absent_score)))
        keep = torch.ones(self.n_classes, dtype=torch.bool, device=scores.device)
        if self.ignore_index is not None and 0 <= self.ignore_index < self.n_classes:
            keep[self.ignore_index] = False
        scores = scores[keep]
        return {"none": lambda s: s, "sum": torch.sum, "elementwise_mean": torch.mean}[self.reduction](scores)


class PanopticMetric(_Accumulator):
    def __init__(self, n_classes: int, temporally_consistent: bool = True, vehicles_id: int = 1, compute_on_step: bool = False):
        super().__init__(("iou", "true_positive", "false_positive", "false_negative"), n_classes)
        self.keys = list(self._counter_names)
        self.n_classes, self.temporally_consistent, self.vehicles_id = n_classes, temporally_consistent, vehicles_id

    def update(self, pred_instance, gt_instance):
        """pred_instance, gt_instance: [batch, time, H, W] instance ids (0 = background)."""
        runtime.require_cuda(pred_instance, gt_instance)
        if pred_instance.shape != gt_instance.shape or gt_instance.dim() != 4:
            raise ValueError("expected two [batch, time, H, W] id tensors of the same shape")
        batch, steps = gt_instance.shape[:2]
        top = torch.stack([pred_instance.max(), gt_instance.max(), -gt_instance.min()]).cpu()
        assert int(top[2]) == 0, "ID 0 of gt_instance must be background"
        K = int(max(top[0], top[1])) + 1
        hist, flag = frame_confusions(pred_instance, gt_instance, batch * steps, K)
        hist = hist.cpu().numpy().reshape(batch, steps, K, K)          # the one device -> host copy of this update
        if int(flag.item()):
            raise ValueError("negative instance id")
        tally = {k: np.zeros(self.n_classes, dtype=np.float64) for k in self.keys}
        for b in range(batch):
            partner = {}            # ground-truth vehicle id -> predicted id of its latest match (id switches)
            for t in range(steps):
                self._score_frame(hist[b, t], partner, tally)
        dev = pred_instance.device
        for k in self.keys:
            self._accumulate(k, torch.from_numpy(tally[k]).to(dev))

    def _score_frame(self, joint, partner, tally):
        """joint[g, p]: pixels with ground-truth id g and predicted id p in one frame."""
        area_gt, area_pred = joint.sum(1), joint.sum(0)
        union = (area_gt[:, None] + area_pred[None, :] - joint).astype(np.float32)
        iou = np.where(union > 0, (joint.astype(np.float32) + np.float32(1e-9)) / (union + np.float32(1e-9)), np.float32(0))
        hits = iou > 0.5
        if hits[0, 0]:                                   # the background segment is scored as a class-0 match only
            tally["true_positive"][0] += 1
            tally["iou"][0] += iou[0, 0]
        gt_ids, pred_ids = np.nonzero(hits[1:, 1:])      # thing <-> thing matches, ascending ground-truth id
        gt_ids, pred_ids = gt_ids + 1, pred_ids + 1
        track = self.temporally_consistent and THING_CLASS == self.vehicles_id
        for g, p in zip(gt_ids.tolist(), pred_ids.tolist()):
            switched = track and g in partner and partner[g] != p
            partner[g] = p
            if switched:
                tally["false_negative"][THING_CLASS] += 1
                tally["false_positive"][THING_CLASS] += 1
            else:
                tally["true_positive"][THING_CLASS] += 1
                tally["iou"][THING_CLASS] += iou[g, p]
        present_gt, present_pred = area_gt[1:] > 0, area_pred[1:] > 0
        matched_gt, matched_pred = hits[1:, 1:].any(1), hits[1:, 1:].any(0)
        tally["false_negative"][THING_CLASS] += int(np.count_nonzero(present_gt & ~matched_gt))
        tally["false_positive"][THING_CLASS] += int(np.count_nonzero(present_pred & ~matched_pred))

    def compute(self):
        one = torch.ones_like(self.true_positive)
        denominator = torch.maximum(self.true_positive + 0.5 * self.false_positive + 0.5 * self.false_negative, one)
        return {"pq": self.iou / denominator, "sq": self.iou / torch.maximum(self.true_positive, one),
                "rq": self.true_positive / denominator}
