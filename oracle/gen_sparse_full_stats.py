"""Test infrastructure (CPU): statistics of the SparseEncoder ORACLE on full-size clouds.

5 frames x 350 000 x 5 points (tools/voxelbench.cloud, seeds 10..14) -> hard voxelisation (oracle/voxelize.py, shipped grid
1600 x 1600 x 40, <= 10 points / voxel, <= 160 000 voxels per cloud) -> per-voxel mean -> oracle/sparse_encoder_ref.py with the
shipped channel widths and the hashed weights of tests/test_gpu_end_to_end.py::_shipped_model -> BEV [5, 256, 200, 200].
The tensor is 205 MB, so only statistics are committed (tests/golden/sparse_full_cloud_stats.json): mean, mean-abs, abs-max,
occupancy, float64 sum, and 512 strided samples per frame.  The GPU test compares the product's extract_lidar_features on the
same clouds.  The sparse oracle itself is "parity unpinned" (spconv cannot be built here): this pins the PRODUCT to
the restatement at full size, nothing more.  Usage: python3 oracle/gen_sparse_full_stats.py  (~10 min, ~20 GB)"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

N_SAMPLES = 512


def sample_index(numel, n=N_SAMPLES):
    """n deterministic positions of a flattened frame (odd stride: visits every residue class)"""
    stride = (numel // n) | 1
    return (np.arange(n, dtype=np.int64) * stride + 12345) % numel


def stats_of(lid):
    """lid: [T, C, H, W] float tensor (CPU) -> dict of per-frame statistics"""
    out = []
    for t in range(lid.shape[0]):
        f = lid[t].double().flatten()
        idx = torch.from_numpy(sample_index(f.numel()))
        out.append({"mean": float(f.mean()), "mean_abs": float(f.abs().mean()), "abs_max": float(f.abs().max()),
                    "occupancy": float((lid[t].abs().amax(0) > 0).double().mean()), "sum": float(f.sum()),
                    "samples": [float(v) for v in lid[t].flatten()[idx]]})
    return out


def main():
    import voxelbench
    from workloads import hashfill
    from oracle import sparse_encoder_ref as SR, voxelize as VZ
    from streamingflow_amd.models import streamingflow as SFM
    torch.set_num_threads(8)
    cfg = SFM.default_cfg()
    net = SFM.streamingflow(cfg).eval()
    pb = "encoders.lidar.backbone."
    sd = hashfill.fill_state_dict({k: v for k, v in net.state_dict().items() if k.startswith(pb)}, seed=92, gain=1.6)
    sd = {k[len(pb):]: v for k, v in sd.items()}
    lidar = SFM.LIDAR_ENCODER
    vz = lidar["voxelize"]
    pts = [voxelbench.cloud(seed=10 + t) for t in range(5)]
    t0 = time.time()
    f, c, _ = VZ.sf_voxelize([p.numpy() for p in pts], vz["voxel_size"], vz["point_cloud_range"], vz["max_num_points"], vz["max_voxels"][1])
    print("voxelised", tuple(f.shape), time.time() - t0, flush=True)
    with torch.no_grad():
        lid = SR.sparse_encoder_forward(sd, f.numpy(), c.numpy(), 5, dict(lidar["backbone"]))
    print("encoded", tuple(lid.shape), time.time() - t0, flush=True)
    out = {"what": __doc__.split("Usage")[0], "clouds": "tools/voxelbench.cloud(seed=10..14), 350000 x 5 each", "voxels": int(f.shape[0]),
           "shape": list(lid.shape), "n_samples": N_SAMPLES, "frames": stats_of(lid)}
    json.dump(out, open(os.path.join(ROOT, "tests", "golden", "sparse_full_cloud_stats.json"), "w"))
    print("written", time.time() - t0)


if __name__ == "__main__":
    main()
