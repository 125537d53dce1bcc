"""Evaluation harness: IoU / panoptic metrics and instance post-processing.
Golden values come from the REFERENCE's own code run here (tests/golden/eval.npz, `gen_golden --only eval`:
streamingflow/utils/instance.py as is, streamingflow/metrics.py on a restated pytorch-lightning Metric base).
Integer / index work: instance maps and counters must match exactly."""
import numpy as np
import pytest
import torch

from util import cases, gold

SEEDS = (0, 1, 2)


def test_golden_is_selfconsistent():
    G = gold("eval.npz")
    for seed in SEEDS:
        inst = G[f"consistent_{seed}"]
        assert inst.shape == (1, 4, 48, 40) and inst.max() >= 3
        assert 0.0 <= float(G[f"iou_{seed}"][1]) <= 1.0 and G[f"pq_{seed}"].shape == (2,)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", SEEDS)
def test_instance_postprocessing_and_metrics(seed):
    from streamingflow_amd import instance as I
    from streamingflow_amd.metrics import IntersectionOverUnion, PanopticMetric
    G = gold("eval.npz")
    out, labels = cases.eval_scene(seed)
    dout = {k: v.cuda() for k, v in out.items()}
    # single frame pieces
    centers = I.find_instance_centers(dout["instance_center"][0, 0], conf_threshold=0.1)
    assert np.array_equal(centers.cpu().numpy(), G[f"centers_{seed}"])
    fg = torch.argmax(dout["segmentation"][0, 0], 0) == 1
    inst0, c0 = I.get_instance_segmentation_and_centers(dout["instance_center"][0, 0], dout["instance_offset"][0, 0], fg)
    assert np.array_equal(inst0.cpu().numpy(), G[f"inst0_{seed}"])
    # whole sequence
    cons = I.predict_instance_segmentation_and_trajectories(dout, compute_matched_centers=False, make_consistent=True)
    assert cons.dtype == torch.int64 and np.array_equal(cons.cpu().numpy(), G[f"consistent_{seed}"])
    cons2, traj = I.predict_instance_segmentation_and_trajectories({k: v.clone() for k, v in dout.items()}, compute_matched_centers=True)
    assert torch.equal(cons2, cons) and len(traj) >= 3
    for k, v in traj.items():
        assert np.allclose(v, G[f"traj_{seed}_{k}"], atol=1e-4)
    # metrics
    iou = IntersectionOverUnion(2).cuda()
    seg_pred = torch.argmax(dout["segmentation"], dim=2, keepdim=True)
    iou(seg_pred, labels["segmentation"].cuda())
    iou(seg_pred[:, 1:], labels["segmentation"].cuda()[:, 1:])
    assert np.array_equal(iou.true_positive.cpu().numpy(), G[f"iou_tp_{seed}"])
    assert np.array_equal(iou.false_positive.cpu().numpy(), G[f"iou_fp_{seed}"])
    assert np.array_equal(iou.false_negative.cpu().numpy(), G[f"iou_fn_{seed}"])
    assert np.allclose(iou.compute().cpu().numpy(), G[f"iou_{seed}"], atol=1e-7)
    pq = PanopticMetric(2).cuda()
    pq(cons, labels["instance"].cuda())
    res = pq.compute()
    for k in ("true_positive", "false_positive", "false_negative"):
        assert np.array_equal(getattr(pq, k).cpu().numpy(), G[f"pq_{k}_{seed}"]), k
    assert np.allclose(pq.iou.cpu().numpy(), G[f"pq_iou_{seed}"], atol=1e-5)
    assert np.allclose(res["pq"].cpu().numpy(), G[f"pq_{seed}"], atol=1e-5)
    pq.reset()
    assert float(pq.true_positive.sum()) == 0.0


@pytest.mark.gpu
def test_confusion_rejects_out_of_range_and_cpu():
    from streamingflow_amd.metrics import IntersectionOverUnion, confusion
    a = torch.tensor([0, 1, 2, 1], device="cuda")
    conf, bad = confusion(a, torch.tensor([0, 1, 1, 1], device="cuda"), 3)
    assert conf.cpu().tolist() == [[1, 0, 0], [0, 2, 1], [0, 0, 0]] and int(bad.item()) == 0
    _, bad = confusion(a, a, 2)
    assert int(bad.item()) == 1
    with pytest.raises(RuntimeError):
        IntersectionOverUnion(2)(torch.zeros(4, dtype=torch.long), torch.zeros(4, dtype=torch.long))
