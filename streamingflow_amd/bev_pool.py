"""``bev_pool(feats, coords, B, D, H, W)`` — MI355X-native drop-in for ``mmdet3d.ops.bev_pool``
(mmdet3d/ops/bev_pool/bev_pool.py:85-98): sum the feature rows that share an integer coordinate.

The reference builds int64 ranks, argsorts them, gathers feats/coords into sorted order, derives
interval tables with boolean indexing and calls a CUDA kernel.  Here one library call ranks and
sorts (cell id, point id) pairs on the device (stable: points of a cell are summed in ascending
point index — the reference leaves the order of equal ranks to ``argsort``), a second one sums
every cell straight from the unsorted feature matrix.  No host synchronisation, no CPU fallback.
"""
import torch

from . import _lib, runtime
from .runtime import ptr


def cell_index_from_coords(coords, B, Z, X, Y):
    """coords [n, 4] = (x, y, z, b) integer -> (order [n] int32, cell_start [B*Z*X*Y + 1] int32)."""
    runtime.require_cuda(coords)
    dev = coords.device
    c32 = coords.to(torch.int32).contiguous()
    n = c32.shape[0]
    ncells = B * Z * X * Y
    order = torch.empty((max(n, 1),), dtype=torch.int32, device=dev)
    start = torch.zeros((ncells + 1,), dtype=torch.int32, device=dev)
    if n:
        L = _lib.lib()
        ws = runtime.workspace(L.sf_lift_index_ws_bytes(n, ncells), dev)
        _lib.check(L.sf_lift_index_coords_fwd(ptr(c32), n, B, Z, X, Y, ptr(order), ptr(start), ptr(ws), ws.numel() * 4,
                                              runtime.stream_ptr(dev)), "lift_index_coords")
    return order, start


def pool_cells(x, order, cell_start, n_cells, prev=None, discount=1.0, out=None):
    """x [n_points, C] fp32 -> [n_cells, C]: per-cell sums (+ prev * discount when prev is given)."""
    C = x.shape[1]
    if out is None:
        out = torch.empty((n_cells, C), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().sf_lift_pool_fwd(ptr(x), ptr(order), ptr(cell_start), n_cells, C, ptr(prev), float(discount), ptr(out),
                                           runtime.stream_ptr(x.device)), "lift_pool")
    return out


def bev_pool(feats, coords, B, D, H, W):
    """feats [n, C] fp32, coords [n, 4] = (x, y, z, b) -> [B, C, D, H, W] (bev_pool.py:85-98)."""
    assert feats.shape[0] == coords.shape[0]
    runtime.require_cuda(feats, coords)
    B, D, H, W = int(B), int(D), int(H), int(W)
    x = runtime.f32c(feats)
    C = x.shape[1]
    if x.shape[0] == 0:
        return torch.zeros((B, C, D, H, W), dtype=torch.float32, device=x.device)
    order, start = cell_index_from_coords(coords, B, D, H, W)
    out = pool_cells(x, order, start, B * D * H * W)
    return out.view(B, D, H, W, C).permute(0, 4, 1, 2, 3).contiguous()


def bev_pool_forward(x, geom_feats, interval_lengths, interval_starts, B, D, H, W):
    """``bev_pool_ext.bev_pool_forward`` (bev_pool.cpp:26-49): pre-sorted x [n, C], int32 geom_feats
    [n, 4], one interval per occupied cell -> [B, D, H, W, C]."""
    runtime.require_cuda(x, geom_feats, interval_lengths, interval_starts)
    x = runtime.f32c(x)
    g = geom_feats.to(torch.int32).contiguous()
    ln, st = interval_lengths.to(torch.int32).contiguous(), interval_starts.to(torch.int32).contiguous()
    out = torch.empty((int(B), int(D), int(H), int(W), x.shape[1]), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().sf_bev_pool_fwd(ptr(x), ptr(g), ptr(ln), ptr(st), x.shape[0], x.shape[1], st.shape[0], int(B), int(D),
                                          int(H), int(W), ptr(out), runtime.stream_ptr(x.device)), "bev_pool")
    return out
