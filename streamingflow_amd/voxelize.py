"""LiDAR hard voxelisation — MI355X-native drop-in for ``mmdet3d.ops.voxel.Voxelization``
(mmdet3d/ops/voxel/voxelize.py:75-139) and for the per-sample loop + mean reduction of
``streamingflow.voxelize`` (streamingflow/models/streamingflow.py:170-198).

One library call per point cloud (key, stable radix sort, scan, assign — csrc/voxelize.hip) instead of
the reference's O(N^2) duplicate search, single-thread numbering kernel and device synchronisations.
Results are bit-identical to the reference's deterministic implementation.  No CPU fallback.
"""
import ctypes as C

import torch
from torch import nn
from torch.nn.modules.utils import _pair

from . import _lib, runtime
from .runtime import ptr


def hard_voxelize_padded(points, voxel_size, coors_range, max_points, max_voxels, want_voxels=True, want_mean=False):
    """-> dict(voxels [max_voxels, max_points, F] | None, coors [max_voxels, 3] int32, num [max_voxels] int32,
    mean [max_voxels, F] | None, voxel_num: 0-d int32 device tensor) — everything stays on the device."""
    runtime.require_cuda(points)
    pts = runtime.f32c(points)
    n, F = pts.shape
    dev = pts.device
    max_points, max_voxels = int(max_points), int(max_voxels)
    voxels = torch.empty((max_voxels, max_points, F), dtype=torch.float32, device=dev) if want_voxels else None
    coors = torch.empty((max_voxels, 3), dtype=torch.int32, device=dev)
    num = torch.empty((max_voxels,), dtype=torch.int32, device=dev)
    mean = torch.empty((max_voxels, F), dtype=torch.float32, device=dev) if want_mean else None
    vnum = torch.empty((), dtype=torch.int32, device=dev)
    L = _lib.lib()
    ws = runtime.workspace(L.sf_hard_voxelize_ws_bytes(max(n, 1)), dev)
    vs = (C.c_float * 3)(*[float(v) for v in voxel_size])
    cr = (C.c_float * 6)(*[float(v) for v in coors_range])
    _lib.check(L.sf_hard_voxelize_fwd(ptr(pts), n, F, vs, cr, max_points, max_voxels, ptr(voxels), ptr(coors), ptr(num), ptr(mean),
                                      ptr(vnum), ptr(ws), ws.numel() * 4, runtime.stream_ptr(dev)), "hard_voxelize")
    return {"voxels": voxels, "coors": coors, "num": num, "mean": mean, "voxel_num": vnum}


def dynamic_voxelize(points, voxel_size, coors_range):
    """[N, F>=3] points -> coors [N, 3] int32: the (x, y, z) voxel of every point, (-1, -1, -1) outside the range
    (voxelize.py:46-49; mmdet3d/ops/voxel/src/voxelization_cpu.cpp:8-43)."""
    runtime.require_cuda(points)
    pts = points.contiguous().float()
    n, F = pts.shape
    coors = torch.empty((n, 3), dtype=torch.int32, device=pts.device)
    vs = (C.c_float * 3)(*[float(v) for v in voxel_size])
    cr = (C.c_float * 6)(*[float(v) for v in coors_range])
    _lib.check(_lib.lib().sf_dynamic_voxelize_fwd(ptr(pts), n, F, vs, cr, ptr(coors), runtime.stream_ptr(pts.device)), "dynamic_voxelize")
    return coors


class Voxelization(nn.Module):
    """Same constructor, attributes and ``forward`` as the reference module (voxelize.py:75-139)."""

    def __init__(self, voxel_size, point_cloud_range, max_num_points, max_voxels=20000, deterministic=True):
        super().__init__()
        self.voxel_size = voxel_size
        self.point_cloud_range = point_cloud_range
        self.max_num_points = max_num_points
        self.max_voxels = max_voxels if isinstance(max_voxels, tuple) else _pair(max_voxels)
        self.deterministic = deterministic      # this implementation is always deterministic
        pcr = torch.tensor(point_cloud_range, dtype=torch.float32)
        vs = torch.tensor(voxel_size, dtype=torch.float32)
        grid_size = torch.round((pcr[3:] - pcr[:3]) / vs).long()
        self.grid_size = grid_size
        self.pcd_shape = [*grid_size[:2], 1]

    def forward(self, input):
        """input [N, F>=3] -> (voxels [M, max_points, F], coors [M, 3] int32 (x, y, z), num_points_per_voxel [M])."""
        max_voxels = self.max_voxels[0] if self.training else self.max_voxels[1]
        if self.max_num_points == -1 or max_voxels == -1:      # voxelize.py:46-49: the coordinates only
            return dynamic_voxelize(input, self.voxel_size, self.point_cloud_range)
        r = hard_voxelize_padded(input, self.voxel_size, self.point_cloud_range, self.max_num_points, max_voxels)
        m = int(r["voxel_num"].item())        # the reference slices too (voxelize.py:68-71): one host sync
        return r["voxels"][:m], r["coors"][:m], r["num"][:m]

    def __repr__(self):
        return (f"{self.__class__.__name__}(voxel_size={self.voxel_size}, point_cloud_range={self.point_cloud_range}, "
                f"max_num_points={self.max_num_points}, max_voxels={self.max_voxels}, deterministic={self.deterministic})")


def voxelize(points, voxelizer, voxelize_reduce=True):
    """``streamingflow.voxelize`` (streamingflow.py:170-198): list of per-sample point clouds ->
    (feats [M, F] mean point of every voxel, coords [M, 4] int32 = (sample, x, y, z), sizes [M] int32).
    The [M, max_points, F] tensor is never materialised when ``voxelize_reduce`` is set."""
    feats, coords, sizes = [], [], []
    max_voxels = voxelizer.max_voxels[0] if voxelizer.training else voxelizer.max_voxels[1]
    rs = [hard_voxelize_padded(res, voxelizer.voxel_size, voxelizer.point_cloud_range, voxelizer.max_num_points, max_voxels,
                               want_voxels=not voxelize_reduce, want_mean=voxelize_reduce) for res in points]
    counts = torch.stack([r["voxel_num"] for r in rs]).tolist()      # one host sync for the whole batch
    for k, (r, m) in enumerate(zip(rs, counts)):
        feats.append((r["mean"] if voxelize_reduce else r["voxels"])[:m])
        coords.append(torch.nn.functional.pad(r["coors"][:m], (1, 0), mode="constant", value=k))
        sizes.append(r["num"][:m])
    feats, coords, sizes = torch.cat(feats, 0), torch.cat(coords, 0), torch.cat(sizes, 0)
    return feats.contiguous(), coords, sizes
