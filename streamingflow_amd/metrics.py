"""Evaluation metrics of StreamingFlow on the MI355X: ``IntersectionOverUnion`` and
``PanopticMetric`` with the reference's constructor arguments, ``update`` / ``compute`` / ``__call__`` and state
names (streamingflow/metrics.py:15-261), without ``pytorch_lightning.metrics`` (removed upstream).

The per-pixel work — the joint label histogram behind the IoU statistics and behind PanopticMetric's
``bincount(prediction + K * target)`` — is one integer kernel (``sf_confusion_fwd``: exact); the matching
logic on the K x K matrix (a few dozen entries) stays on the host as in the reference.  States are plain
tensors; ``streamingflow_amd.dist.reduce_counters`` sums them across ranks (the reference declares
``dist_reduce_fx='sum'``).  CUDA tensors only.
"""
from typing import Optional

import torch
import torch.nn as nn

from . import _lib, runtime
from .runtime import ptr


def confusion(pred, target, K):
    """[K, K] int64, entry [t, p] = number of positions with target t and prediction p (labels must be in [0, K))."""
    runtime.require_cuda(pred, target)
    a = pred.reshape(-1).to(torch.int64).contiguous()
    b = target.reshape(-1).to(torch.int64).contiguous()
    if a.numel() != b.numel():
        raise ValueError("prediction and target must have the same number of elements")
    out = torch.empty((K, K), dtype=torch.int64, device=a.device)
    bad = torch.empty((1,), dtype=torch.int32, device=a.device)
    _lib.check(_lib.lib().sf_confusion_fwd(ptr(a), ptr(b), a.numel(), int(K), ptr(out), ptr(bad), runtime.stream_ptr(a.device)), "confusion")
    return out, bad


class _Metric(nn.Module):
    def __init__(self):
        super().__init__()
        self._state_defaults = {}

    def add_state(self, name, default, dist_reduce_fx="sum"):
        self._state_defaults[name] = default.clone()
        self.register_buffer(name, default.clone(), persistent=False)

    def reset(self):
        for k, v in self._state_defaults.items():
            getattr(self, k).copy_(v)

    def forward(self, *args, **kwargs):
        self.update(*args, **kwargs)

    def sync(self):
        """Sum the states over the ranks of the default process group (no-op when not initialised)."""
        from . import dist
        for k in self._state_defaults:
            dist.reduce_counters(getattr(self, k))


class IntersectionOverUnion(_Metric):
    """metrics.py:15-71."""

    def __init__(self, n_classes: int, ignore_index: Optional[int] = None, absent_score: float = 0.0, reduction: str = "none",
                 compute_on_step: bool = False):
        super().__init__()
        self.n_classes, self.ignore_index, self.absent_score, self.reduction = n_classes, ignore_index, absent_score, reduction
        for name in ("true_positive", "false_positive", "false_negative", "support"):
            self.add_state(name, torch.zeros(n_classes))

    def update(self, prediction: torch.Tensor, target: torch.Tensor):
        if prediction.ndim == target.ndim + 1:      # stat_scores_multiple_classes' argmax_dim=1 convention
            prediction = torch.argmax(prediction, dim=1)
        K = self.n_classes + 1                      # the extra bin pytorch-lightning keeps for an ignored label
        conf, bad = confusion(prediction, target, K)
        conf = conf.to(torch.float32)
        diag = conf.diagonal()
        self.true_positive += diag[: self.n_classes].to(self.true_positive.device)
        self.false_positive += (conf.sum(0) - diag)[: self.n_classes]
        self.false_negative += (conf.sum(1) - diag)[: self.n_classes]
        self.support += conf.sum(1)[: self.n_classes]
        self._bad = bad

    def compute(self):
        if getattr(self, "_bad", None) is not None and int(self._bad.item()):
            raise RuntimeError("IntersectionOverUnion: a label outside [0, n_classes] was seen")
        scores = torch.zeros(self.n_classes, device=self.true_positive.device, dtype=torch.float32)
        for c in range(self.n_classes):
            if c == self.ignore_index:
                continue
            tp, fp, fn, sup = self.true_positive[c], self.false_positive[c], self.false_negative[c], self.support[c]
            if sup + tp + fp == 0:
                scores[c] = self.absent_score
                continue
            scores[c] = tp.to(torch.float) / (tp + fp + fn)
        if (self.ignore_index is not None) and (0 <= self.ignore_index < self.n_classes):
            scores = torch.cat([scores[: self.ignore_index], scores[self.ignore_index + 1:]])
        if self.reduction == "elementwise_mean":
            return torch.mean(scores)
        if self.reduction == "sum":
            return torch.sum(scores)
        if self.reduction == "none":
            return scores
        raise ValueError("Reduction parameter unknown.")


class PanopticMetric(_Metric):
    """metrics.py:74-261."""

    def __init__(self, n_classes: int, temporally_consistent: bool = True, vehicles_id: int = 1, compute_on_step: bool = False):
        super().__init__()
        self.n_classes, self.temporally_consistent, self.vehicles_id = n_classes, temporally_consistent, vehicles_id
        self.keys = ["iou", "true_positive", "false_positive", "false_negative"]
        for k in self.keys:
            self.add_state(k, torch.zeros(n_classes))

    def update(self, pred_instance, gt_instance):
        runtime.require_cuda(pred_instance, gt_instance)
        batch_size, sequence_length = gt_instance.shape[:2]
        assert gt_instance.min() == 0, "ID 0 of gt_instance must be background"
        pred_segmentation = (pred_instance > 0).long()
        gt_segmentation = (gt_instance > 0).long()
        for b in range(batch_size):
            unique_id_mapping = {}
            for t in range(sequence_length):
                result = self.panoptic_metrics(pred_segmentation[b, t].detach(), pred_instance[b, t].detach(), gt_segmentation[b, t],
                                               gt_instance[b, t], unique_id_mapping)
                for k in self.keys:
                    getattr(self, k).add_(result[k].to(getattr(self, k).device))

    def compute(self):
        denominator = torch.maximum(self.true_positive + self.false_positive / 2 + self.false_negative / 2,
                                    torch.ones_like(self.true_positive))
        pq = self.iou / denominator
        sq = self.iou / torch.maximum(self.true_positive, torch.ones_like(self.true_positive))
        rq = self.true_positive / denominator
        return {"pq": pq, "sq": sq, "rq": rq}

    def panoptic_metrics(self, pred_segmentation, pred_instance, gt_segmentation, gt_instance, unique_id_mapping):
        """metrics.py:143-226: the joint histogram runs on the device, the (small) matching on the host."""
        n_classes = self.n_classes
        assert pred_segmentation.dim() == 2
        assert pred_segmentation.shape == pred_instance.shape == gt_segmentation.shape == gt_instance.shape
        n_instances = int(torch.cat([pred_instance, gt_instance]).max().item())
        n_all_things = n_instances + n_classes
        n_things_and_void = n_all_things + 1
        prediction, pred_to_cls = self.combine_mask(pred_segmentation, pred_instance, n_classes, n_all_things)
        target, target_to_cls = self.combine_mask(gt_segmentation, gt_instance, n_classes, n_all_things)
        conf, bad = confusion(prediction, target, n_things_and_void)        # [target][prediction]
        conf = conf.cpu()
        if int(bad.item()):
            raise ValueError("Incorrect bincount size.")
        pred_to_cls, target_to_cls = pred_to_cls.cpu(), target_to_cls.cpu()
        result = {key: torch.zeros(n_classes, dtype=torch.float32) for key in self.keys}
        conf = conf[1:, 1:]                                                  # drop the void class
        union = conf.sum(0).unsqueeze(0) + conf.sum(1).unsqueeze(1) - conf
        iou = torch.where(union > 0, (conf.float() + 1e-9) / (union.float() + 1e-9), torch.zeros_like(union).float())
        mapping = (iou > 0.5).nonzero(as_tuple=False)
        is_matching = pred_to_cls[mapping[:, 1]] == target_to_cls[mapping[:, 0]]
        mapping = mapping[is_matching]
        tp_mask = torch.zeros_like(conf, dtype=torch.bool)
        tp_mask[mapping[:, 0], mapping[:, 1]] = True
        for target_id, pred_id in mapping:
            cls_id = pred_to_cls[pred_id]
            if self.temporally_consistent and cls_id == self.vehicles_id:
                if target_id.item() in unique_id_mapping and unique_id_mapping[target_id.item()] != pred_id.item():
                    result["false_negative"][target_to_cls[target_id]] += 1
                    result["false_positive"][pred_to_cls[pred_id]] += 1
                    unique_id_mapping[target_id.item()] = pred_id.item()
                    continue
            result["true_positive"][cls_id] += 1
            result["iou"][cls_id] += iou[target_id][pred_id]
            unique_id_mapping[target_id.item()] = pred_id.item()
        for target_id in range(n_classes, n_all_things):
            if tp_mask[target_id, n_classes:].any():
                continue
            if target_to_cls[target_id] != -1:
                result["false_negative"][target_to_cls[target_id]] += 1
        for pred_id in range(n_classes, n_all_things):
            if tp_mask[n_classes:, pred_id].any():
                continue
            if pred_to_cls[pred_id] != -1 and (conf[:, pred_id] > 0).any():
                result["false_positive"][pred_to_cls[pred_id]] += 1
        return result

    def combine_mask(self, segmentation, instance, n_classes, n_all_things):
        """metrics.py:228-261 (tensor plumbing, unchanged semantics)."""
        instance = instance.view(-1)
        instance_mask = instance > 0
        instance = instance - 1 + n_classes
        segmentation = segmentation.clone().view(-1)
        segmentation_mask = segmentation < n_classes
        keep = instance_mask & segmentation_mask
        tuples = torch.cat((instance[keep].unsqueeze(1), segmentation[keep].unsqueeze(1)), dim=1)
        instance_id_to_class = -tuples.new_ones((n_all_things,))
        instance_id_to_class[tuples[:, 0]] = tuples[:, 1]
        instance_id_to_class[torch.arange(n_classes, device=segmentation.device)] = torch.arange(n_classes, device=segmentation.device)
        segmentation[instance_mask] = instance[instance_mask]
        segmentation += 1
        segmentation[~segmentation_mask] = 0
        return segmentation, instance_id_to_class
