"""TEST INFRASTRUCTURE — CPU restatement of the camera lift-splat voxel pooling.

Not imported by the product.  Every function cites the reference lines it follows; the only piece
of the reference that cannot be executed here is the CUDA kernel of ``bev_pool_ext`` (45 lines,
``mmdet3d/ops/bev_pool/src/bev_pool_cuda.cu``), restated in ``bev_pool_kernel``; the Python around
it (``bev_pool.py``, ``streamingflow.bev_pool`` / ``projection_to_birds_eye_view`` /
``get_geometry`` / ``create_frustum``) is executed from /root/reference by ``oracle/gen_golden.py
--only lift`` to pin this file (see ``refimport.lift_splat_reference``).

Parity status: pinned against the reference's Python (run here) and against the reference's own
second implementation of the pooling (``QuickCumsum``, bev_pool.py:8-33); the CUDA kernel itself is
unbuildable here (no nvcc), its restatement is cross-checked against ``QuickCumsum``.
"""
import numpy as np
import torch


# ---- streamingflow/utils/geometry.py -------------------------------------------------------------
def calculate_birds_eye_view_parameters(x_bounds, y_bounds, z_bounds):
    """geometry.py:40-59 -> (resolution f32[3], start_position f32[3] (first cell centre), dimension i64[3])."""
    rows = [x_bounds, y_bounds, z_bounds]
    res = torch.tensor([r[2] for r in rows])
    start = torch.tensor([r[0] + r[2] / 2.0 for r in rows])
    dim = torch.tensor([(r[1] - r[0]) / r[2] for r in rows], dtype=torch.long)
    return res, start, dim


def euler2mat(angle):
    """geometry.py:124-155: R = Rx(x) @ Ry(y) @ Rz(z) from [..., 3] angles."""
    shape = angle.shape
    a = angle.reshape(-1, 3)
    x, y, z = a[:, 0], a[:, 1], a[:, 2]
    zeros, ones = torch.zeros_like(z), torch.ones_like(z)
    cz, sz = torch.cos(z), torch.sin(z)
    zmat = torch.stack([cz, -sz, zeros, sz, cz, zeros, zeros, zeros, ones], dim=1).view(-1, 3, 3)
    cy, sy = torch.cos(y), torch.sin(y)
    ymat = torch.stack([cy, zeros, sy, zeros, ones, zeros, -sy, zeros, cy], dim=1).view(-1, 3, 3)
    cx, sx = torch.cos(x), torch.sin(x)
    xmat = torch.stack([ones, zeros, zeros, zeros, cx, -sx, zeros, sx, cx], dim=1).view(-1, 3, 3)
    return xmat.bmm(ymat).bmm(zmat).view(*shape[:-1], 3, 3)


def pose_vec2mat(vec):
    """geometry.py:158-172: (tx, ty, tz, rx, ry, rz) -> [..., 4, 4]."""
    t = vec[..., :3].unsqueeze(-1)
    rot = euler2mat(vec[..., 3:].contiguous())
    m = torch.cat([rot, t], dim=-1)
    m = torch.nn.functional.pad(m, [0, 0, 0, 1], value=0)
    m[..., 3, 3] = 1.0
    return m


# ---- streamingflow/models/streamingflow.py -------------------------------------------------------
def create_frustum(final_dim, downsample, d_bound):
    """streamingflow.py:149-168 -> [D, fH, fW, 3] (pixel x, pixel y, depth)."""
    h, w = final_dim
    fh, fw = h // downsample, w // downsample
    depth = torch.arange(*d_bound, dtype=torch.float).view(-1, 1, 1).expand(-1, fh, fw)
    D = depth.shape[0]
    xg = torch.linspace(0, w - 1, fw, dtype=torch.float).view(1, 1, fw).expand(D, fh, fw)
    yg = torch.linspace(0, h - 1, fh, dtype=torch.float).view(1, fh, 1).expand(D, fh, fw)
    return torch.stack((xg, yg, depth), -1)


def get_geometry(frustum, intrinsics, extrinsics):
    """streamingflow.py:277-292: ego-frame (x, y, z) of every frustum point -> [B, N, D, fH, fW, 3]."""
    rotation, translation = extrinsics[..., :3, :3], extrinsics[..., :3, 3]
    B, N, _ = translation.shape
    points = frustum.unsqueeze(0).unsqueeze(0).unsqueeze(-1)
    points = torch.cat((points[:, :, :, :, :, :2] * points[:, :, :, :, :, 2:3], points[:, :, :, :, :, 2:3]), 5)
    combined = rotation.matmul(torch.inverse(intrinsics))
    points = combined.view(B, N, 1, 1, 1, 3, 3).matmul(points).squeeze(-1)
    points = points + translation.view(B, N, 1, 1, 1, 3)
    return points


def depth_outer(feat, depth_logits):
    """streamingflow.py:304-312 (USE_DEPTH_DISTRIBUTION): softmax over depth, outer product with the
    features.  feat [bn, C, fH, fW], depth_logits [bn, D, fH, fW] -> [bn, D, fH, fW, C]."""
    prob = depth_logits.softmax(dim=1)
    x = prob.unsqueeze(1) * feat.unsqueeze(2)          # [bn, C, D, fH, fW]
    return x.permute(0, 2, 3, 4, 1)


# ---- mmdet3d/ops/bev_pool ------------------------------------------------------------------------
def bev_pool_kernel_loops(x, geom_feats, interval_lengths, interval_starts, b, d, h, w):
    """bev_pool.cpp:26-49 + bev_pool_cuda.cu:20-42, literally: out[b,d,h,w,c] zero-filled; one
    sequential fp32 sum per (interval, channel), in the given point order, written at the interval's
    first coords.  Pure loops — small cases only."""
    c = x.shape[1]
    out = torch.zeros((b, d, h, w, c), dtype=x.dtype)
    xs = x.numpy()
    g = geom_feats.numpy()
    o = out.numpy()
    st, ln = interval_starts.numpy(), interval_lengths.numpy()
    for i in range(len(st)):
        s0, L = int(st[i]), int(ln[i])
        acc = np.zeros((c,), dtype=np.float32)
        for k in range(L):
            acc = acc + xs[s0 + k]                     # float32 sequential adds
        gx, gy, gz, gb = (int(v) for v in g[s0])
        o[gb, gz, gx, gy] = acc
    return out


def bev_pool_kernel(x, geom_feats, interval_lengths, interval_starts, b, d, h, w):
    """Same arithmetic as ``bev_pool_kernel_loops`` (sequential fp32 adds in point order, starting from
    +0), vectorised with a single-threaded ``index_add_`` so that full-size inputs finish in seconds."""
    n, c = x.shape
    out = torch.zeros((b * d * h * w, c), dtype=x.dtype)
    if n == 0 or interval_starts.numel() == 0:
        return out.view(b, d, h, w, c)
    st = interval_starts.long()
    g0 = geom_feats[st].long()
    dst_of_interval = ((g0[:, 3] * d + g0[:, 2]) * h + g0[:, 0]) * w + g0[:, 1]
    interval_of_point = torch.repeat_interleave(torch.arange(st.numel()), interval_lengths.long())
    nt = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        out.index_add_(0, dst_of_interval[interval_of_point], x)
    finally:
        torch.set_num_threads(nt)
    return out.view(b, d, h, w, c)


def bev_pool_op(feats, coords, B, D, H, W, stable=True):
    """bev_pool.py:85-98 with QuickCumsumCuda.forward (:38-56).  coords [n,4] = (x, y, z, b).
    ``stable`` picks the order of equal ranks (the reference's ``argsort`` leaves it unspecified);
    the HIP path sorts stably, i.e. points of a cell are summed in ascending point index."""
    assert feats.shape[0] == coords.shape[0]
    ranks = coords[:, 0] * (W * D * B) + coords[:, 1] * (D * B) + coords[:, 2] * B + coords[:, 3]
    idx = torch.argsort(ranks, stable=True) if stable else ranks.argsort()
    feats, coords, ranks = feats[idx], coords[idx], ranks[idx]
    kept = torch.ones(feats.shape[0], dtype=torch.bool)
    kept[1:] = ranks[1:] != ranks[:-1]
    starts = torch.where(kept)[0].int()
    lengths = torch.zeros_like(starts)
    if starts.numel():
        lengths[:-1] = starts[1:] - starts[:-1]
        lengths[-1] = feats.shape[0] - starts[-1]
    out = bev_pool_kernel(feats, coords.int(), lengths, starts, B, D, H, W)
    return out.permute(0, 4, 1, 2, 3).contiguous()


def quantise(geom, start, res):
    """streamingflow.py:353: ((g - (start - res/2)) / res).long() — fp32 arithmetic, truncation toward 0."""
    return ((geom - (start - res / 2.0)) / res).long()


def sf_bev_pool(geom_feats, x, start, res, dim, stable=True):
    """streamingflow.bev_pool (streamingflow.py:342-378).  geom_feats [B,N,D,H,W,3] float,
    x [B,N,D,H,W,C] -> (pooled [B, C, Z, X, Y], kept integer coords [n_kept, 4])."""
    B, N, D, H, W, C = x.shape
    Np = B * N * D * H * W
    x = x.reshape(Np, C)
    g = quantise(geom_feats, start, res).view(Np, 3)
    batch_ix = torch.cat([torch.full([Np // B, 1], ix, dtype=torch.long) for ix in range(B)])
    g = torch.cat((g, batch_ix), 1)
    kept = ((g[:, 0] >= 0) & (g[:, 0] < dim[0]) & (g[:, 1] >= 0) & (g[:, 1] < dim[1])
            & (g[:, 2] >= 0) & (g[:, 2] < dim[2]))
    x, g = x[kept], g[kept]
    out = bev_pool_op(x, g, B, int(dim[2]), int(dim[0]), int(dim[1]), stable=stable)
    return out, g


def warp_geometry(geometry_b, rotation_b, translation_b):
    """streamingflow.py:386-396 for one sample: frames 0..t are moved by pose t, for t = 0..s-2, in place
    and cumulatively.  geometry_b [s, n, d, h, w, 3] -> warped copy."""
    geo = geometry_b.clone()
    s = geo.shape[0]
    for t in range(s):
        if t != s - 1:
            tmp = geo[:t + 1]
            tmp = rotation_b[t].view(1, 1, 1, 1, 1, 3, 3).matmul(tmp.unsqueeze(-1)).squeeze(-1)
            tmp = tmp + translation_b[t].view(1, 1, 1, 1, 1, 3)
            geo[:t + 1] = tmp
    return geo


def projection_to_birds_eye_view(x, geometry, future_egomotion, start, res, dim, discount, stable=True):
    """streamingflow.py:380-428.  x [b,s,n,d,h,w,c], geometry [b,s,n,d,h,w,3], future_egomotion [b,s,6]
    -> [b, s, c, X, Y] (Z collapsed: the shipped Z_BOUND gives one slice)."""
    batch, s, n, d, h, w, c = x.shape
    out = torch.zeros((batch, s, c, int(dim[0]), int(dim[1])), dtype=torch.float)
    mat = pose_vec2mat(future_egomotion)
    rotation, translation = mat[..., :3, :3], mat[..., :3, 3]
    for b in range(batch):
        geo = warp_geometry(geometry[b], rotation[b], translation[b])
        bev = torch.zeros((int(dim[2]), int(dim[0]), int(dim[1]), c))
        for t in range(s):
            pooled, _ = sf_bev_pool(geo[t].unsqueeze(0), x[b, t].unsqueeze(0), start, res, dim, stable=stable)
            tmp = pooled[0].permute(1, 2, 3, 0)
            bev = bev * discount + tmp
            out[b, t] = bev.permute((0, 3, 1, 2)).squeeze(0)
    return out
