"""TEST INFRASTRUCTURE — CPU restatement of the LiDAR hard voxelisation (first half).

Not imported by the product.  Follows mmdet3d/ops/voxel/src/voxelization_cuda.cu:25-60 (point -> voxel
coordinate), :101-146 + :148-178 (duplicate search, voxel numbering in first-appearance order,
max_points / max_voxels caps), :62-99 (copy-out), mmdet3d/ops/voxel/voxelize.py:13-72, :122-139
(``Voxelization.forward``) and streamingflow/models/streamingflow.py:170-198 (``voxelize``: per-sample
loop, batch index padded in front of the coordinates, mean over the points of a voxel).

Pinned against the reference's own C++ CPU implementation compiled from its sources where they lie
(``oracle/build_ref.py`` -> ``oracle/_ref/voxel_layer*.so``; voxelization_cpu.cpp:45-102) driven by the
reference's own ``voxelize.py``.  That CPU implementation indexes its [grid_z][grid_y][grid_x] lookup
table with (x, y, z) (voxelization_cpu.cpp:75 — this fork dropped the coordinate reversal), so it is
only memory-safe on cubic grids: the fixtures use cubic grids; the shipped 1600 x 1600 x 40 grid is
covered by this restatement (and by properties) only.
"""
import numpy as np
import torch


def grid_size(voxel_size, coors_range):
    """voxelization_cuda.cu:283-285: round((max - min) / size) per axis, fp32."""
    vs = np.asarray(voxel_size, dtype=np.float32)
    cr = np.asarray(coors_range, dtype=np.float32)
    return [int(np.round((cr[3 + i] - cr[i]) / vs[i])) for i in range(3)]


def point_coors(points, voxel_size, coors_range):
    """voxelization_cuda.cu:25-60: c = floor((p - min) / size) in fp32; any axis outside -> -1 (first column)."""
    p = np.asarray(points, dtype=np.float32)[:, :3]
    vs = np.asarray(voxel_size, dtype=np.float32)
    cr = np.asarray(coors_range, dtype=np.float32)
    g = np.asarray(grid_size(voxel_size, coors_range), dtype=np.int64)
    c = np.floor((p - cr[:3]) / vs).astype(np.int64)
    ok = ((c >= 0) & (c < g)).all(1)
    c[~ok] = -1
    return c, ok


def dynamic_voxelize(points, voxel_size, coors_range):
    """voxelize.py:46-49 -> dynamic_voxelize (voxelization_cpu.cpp:8-43): coors [N, 3] int32, (-1, -1, -1) for points outside the range."""
    c, _ = point_coors(points, voxel_size, coors_range)
    return c.astype(np.int32)


def hard_voxelize(points, voxel_size, coors_range, max_points, max_voxels):
    """-> (voxels [M, max_points, F] f32, coors [M, 3] int32 (x, y, z), num_points_per_voxel [M] int32).
    Voxels are numbered in the order their first point appears; a new voxel after max_voxels is
    dropped with all its points; a voxel keeps its first max_points points (index order)."""
    pts = np.asarray(points, dtype=np.float32)
    n, F = pts.shape
    c, ok = point_coors(pts, voxel_size, coors_range)
    g = grid_size(voxel_size, coors_range)
    key = (c[:, 2] * g[1] + c[:, 1]) * g[0] + c[:, 0]
    idx = np.nonzero(ok)[0]
    if idx.size == 0:
        return np.zeros((0, max_points, F), np.float32), np.zeros((0, 3), np.int32), np.zeros((0,), np.int32)
    k = key[idx]
    order = np.argsort(k, kind="stable")                  # groups by cell, ascending point index inside
    ks, ps = k[order], idx[order]
    head = np.ones(ks.shape[0], dtype=bool)
    head[1:] = ks[1:] != ks[:-1]
    head_pos = np.nonzero(head)[0]
    first_pt = ps[head_pos]                               # smallest point index of each cell
    counts = np.diff(np.append(head_pos, ks.shape[0]))
    vid_of_cell = np.argsort(np.argsort(first_pt, kind="stable"), kind="stable")     # rank by first appearance
    cell_of_sorted = np.cumsum(head) - 1
    rank = np.arange(ks.shape[0]) - head_pos[cell_of_sorted]
    vid = vid_of_cell[cell_of_sorted]
    M = min(int(head_pos.size), int(max_voxels))
    voxels = np.zeros((M, max_points, F), np.float32)
    coors = np.zeros((M, 3), np.int32)
    num = np.zeros((M,), np.int32)
    keep = (vid < M) & (rank < max_points)
    voxels[vid[keep], rank[keep]] = pts[ps[keep]]
    hv = vid_of_cell < M
    coors[vid_of_cell[hv]] = c[first_pt[hv]].astype(np.int32)
    num[vid_of_cell[hv]] = np.minimum(counts[hv], max_points).astype(np.int32)
    return voxels, coors, num


def hard_voxelize_loops(points, voxel_size, coors_range, max_points, max_voxels):
    """The same through the sequential formulation of voxelization_cpu.cpp:45-102 (dict instead of the
    dense lookup table).  Pure loops — small cases only; cross-checks the vectorised version."""
    pts = np.asarray(points, dtype=np.float32)
    c, ok = point_coors(pts, voxel_size, coors_range)
    table, vox, coors, num = {}, [], [], []
    for i in range(pts.shape[0]):
        if not ok[i]:
            continue
        k = tuple(int(v) for v in c[i])
        v = table.get(k, -1)
        if v == -1:
            if len(vox) >= max_voxels:
                continue
            v = len(vox)
            table[k] = v
            vox.append(np.zeros((max_points, pts.shape[1]), np.float32))
            coors.append(k)
            num.append(0)
        if num[v] < max_points:
            vox[v][num[v]] = pts[i]
            num[v] += 1
    if not vox:
        return np.zeros((0, max_points, pts.shape[1]), np.float32), np.zeros((0, 3), np.int32), np.zeros((0,), np.int32)
    return np.stack(vox), np.asarray(coors, np.int32), np.asarray(num, np.int32)


def sf_voxelize(points_list, voxel_size, coors_range, max_points, max_voxels):
    """streamingflow.voxelize (streamingflow.py:170-198) with voxelize_reduce: per sample k hard-voxelise,
    coords <- (k, x, y, z), feats <- sum over the max_points slots / number of points.
    -> (feats [M_total, F] f32, coords [M_total, 4] int32, sizes [M_total] int32)."""
    feats, coords, sizes = [], [], []
    for k, pts in enumerate(points_list):
        f, c, n = hard_voxelize(pts, voxel_size, coors_range, max_points, max_voxels)
        feats.append(torch.from_numpy(f))
        coords.append(torch.nn.functional.pad(torch.from_numpy(c), (1, 0), mode="constant", value=k))
        sizes.append(torch.from_numpy(n))
    feats, coords, sizes = torch.cat(feats, 0), torch.cat(coords, 0), torch.cat(sizes, 0)
    feats = feats.sum(dim=1, keepdim=False) / sizes.type_as(feats).view(-1, 1)
    return feats.contiguous(), coords, sizes
