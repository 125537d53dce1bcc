"""Neighbour-table density of the SparseEncoder on the shipped-size synthetic cloud: the fraction of (output site, kernel tap)
pairs that have an input site — the share of the dense 27-tap implicit GEMM that multiplies real data."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from streamingflow_amd.models import sparse_encoder as SE   # noqa: E402
from streamingflow_amd.models.sparse_encoder import SparseEncoder   # noqa: E402
from streamingflow_amd.voxelize import Voxelization, voxelize   # noqa: E402
from oracle import cases, sparse_encoder_ref as SR   # noqa: E402
from workloads import hashfill   # noqa: E402
import voxelbench   # noqa: E402

dev = torch.device("cuda", 0)
cfg = SR.default_cfg()
m = SparseEncoder(cfg["in_channels"], cfg["sparse_shape"], base_channels=cfg["base_channels"], output_channels=cfg["output_channels"],
                  encoder_channels=cfg["encoder_channels"], encoder_paddings=cfg["encoder_paddings"], block_type="basicblock").eval()
shapes = SR.state_dict_shapes(cfg)
m.load_state_dict(hashfill.fill_state_dict({k: torch.empty(v) if v else torch.tensor(0) for k, v in shapes.items()}, seed=83, gain=1.6))
m = m.to(dev)
vs, rng, mp, mv = cases.VOXEL_SHIPPED
vz = Voxelization(list(vs), list(rng), mp, (120000, mv)).eval()
feats, coords, sizes = voxelize([voxelbench.cloud().to(dev)], vz)
stats = []
orig = m._table


def spy(in_coords, out_coords, batch, shape, k, s, p, subm):
    t = orig(in_coords, out_coords, batch, shape, k, s, p, subm)
    stats.append((tuple(k), tuple(s), bool(subm), int(out_coords.shape[0]), float((t >= 0).float().mean()), int(t.shape[1]) if t.dim() > 1 else 0))
    return t


m._table = spy
with torch.no_grad():
    m(feats, coords, 1, nhwc=True)
tot_dense = tot_live = 0.0
for k, s, subm, n, d, taps in stats:
    print(f"kernel {k} stride {s} subm {subm}: {n:7d} output sites, {taps} taps, {100 * d:5.1f} % of the (site, tap) pairs have an input site")
