"""Label warping: streamingflow_amd.labels vs fixtures made by the reference's own geometry helpers
(tests/golden/labels.npz).  Nearest-neighbour warps of categorical maps can differ on isolated pixels whose source
coordinate falls within fp32 rounding of a .5 boundary (affine_grid's matmul order is a backend detail): at most
0.1 % of the pixels may differ; continuous maps: the same pixels aside, <= 1e-5."""
from types import SimpleNamespace as NS

import numpy as np
import pytest
import torch

from util import cases, gold


def test_fixture_shapes():
    G = gold("labels.npz")
    assert G["0.segmentation"].shape == (2, 7, 1, 64, 64) and G["0.instance"].shape == (2, 7, 64, 64)
    assert G["0.depths"].dtype == np.int64 and G["0.depths"].shape == (2, 3, 2, 4, 6)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", (0, 1))
def test_prepare_future_labels(seed):
    from streamingflow_amd import labels as LB
    G = gold("labels.npz")
    cfg = NS(LIFT=NS(GT_DEPTH=True, D_BOUND=[2.0, 50.0, 1.0]), SEMANTIC_SEG=NS(PEDESTRIAN=NS(ENABLED=False)),
             INSTANCE_SEG=NS(ENABLED=True), INSTANCE_FLOW=NS(ENABLED=True))
    batch = {k: v.cuda() for k, v in cases.label_batch(seed).items()}
    lab = LB.prepare_future_labels(batch, cfg, 3, (50.0, 50.0), 8)
    assert np.array_equal(lab["depths"].cpu().numpy(), G[f"{seed}.depths"])
    for k in ("segmentation", "instance", "centerness", "offset", "flow"):
        a, w = lab[k].cpu().numpy(), G[f"{seed}.{k}"]
        assert a.shape == w.shape and a.dtype == w.dtype, k
        bad = np.abs(a.astype(np.float64) - w.astype(np.float64)) > 1e-5
        assert bad.mean() <= 1e-3, (k, bad.mean())
    x = batch["centerness"][:, 0]
    wb = LB.warp_features(x, batch["future_egomotion"][:, 0], mode="bilinear", spatial_extent=(50.0, 50.0))
    assert float(np.abs(wb.cpu().numpy() - G[f"{seed}.warp_bilinear"]).max()) <= 1e-4
    # identity pose: nothing moves
    z = torch.zeros_like(batch["future_egomotion"][:, 0])
    assert torch.equal(LB.warp_features(x, z, mode="nearest", spatial_extent=(50.0, 50.0)), x)
