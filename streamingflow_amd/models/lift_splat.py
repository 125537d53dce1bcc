"""Camera lift-splat: frustum features -> BEV grid, MI355X-native.

Mirrors the pieces of the reference ``streamingflow`` module that sit between the image encoder
and the temporal model (streamingflow/models/streamingflow.py): ``create_frustum`` (:149-168),
``get_geometry`` (:277-292), the depth (x) feature outer product of ``encoder_forward`` (:304-312),
``bev_pool`` (:342-378) and ``projection_to_birds_eye_view`` (:380-428) — same method names,
argument meaning and return values — plus ``lift_splat``: the whole chain in one pass that never
materialises the [b, s, n, D, fH, fW, C] tensor nor the geometry tensor.

Device side: libsfnative (csrc/lift_splat.hip).  No CPU fallback: CPU tensors raise.
"""
import torch
import torch.nn as nn

from .. import _lib, runtime
from ..bev_pool import pool_cells
from ..runtime import ptr


def calculate_birds_eye_view_parameters(x_bounds, y_bounds, z_bounds):
    """utils/geometry.py:40-59."""
    rows = [x_bounds, y_bounds, z_bounds]
    bev_resolution = torch.tensor([row[2] for row in rows])
    bev_start_position = torch.tensor([row[0] + row[2] / 2.0 for row in rows])
    bev_dimension = torch.tensor([(row[1] - row[0]) / row[2] for row in rows], dtype=torch.long)
    return bev_resolution, bev_start_position, bev_dimension


def euler2mat(angle):
    """utils/geometry.py:124-155."""
    shape = angle.shape
    a = angle.reshape(-1, 3)
    x, y, z = a[:, 0], a[:, 1], a[:, 2]
    zeros, ones = torch.zeros_like(z), torch.ones_like(z)
    cz, sz, cy, sy, cx, sx = torch.cos(z), torch.sin(z), torch.cos(y), torch.sin(y), torch.cos(x), torch.sin(x)
    zmat = torch.stack([cz, -sz, zeros, sz, cz, zeros, zeros, zeros, ones], dim=1).view(-1, 3, 3)
    ymat = torch.stack([cy, zeros, sy, zeros, ones, zeros, -sy, zeros, cy], dim=1).view(-1, 3, 3)
    xmat = torch.stack([ones, zeros, zeros, zeros, cx, -sx, zeros, sx, cx], dim=1).view(-1, 3, 3)
    return xmat.bmm(ymat).bmm(zmat).view(*shape[:-1], 3, 3)


def pose_vec2mat(vec):
    """utils/geometry.py:158-172: (tx, ty, tz, rx, ry, rz) -> [..., 4, 4]."""
    translation = vec[..., :3].unsqueeze(-1)
    rot_mat = euler2mat(vec[..., 3:].contiguous())
    m = torch.cat([rot_mat, translation], dim=-1)
    m = torch.nn.functional.pad(m, [0, 0, 0, 1], value=0)
    m[..., 3, 3] = 1.0
    return m


class LiftSplat(nn.Module):
    def __init__(self, x_bound=(-50.0, 50.0, 0.5), y_bound=(-50.0, 50.0, 0.5), z_bound=(-10.0, 10.0, 20.0),
                 d_bound=(2.0, 50.0, 1.0), final_dim=(224, 480), encoder_downsample=8, discount=0.5):
        super().__init__()
        res, start, dim = calculate_birds_eye_view_parameters(list(x_bound), list(y_bound), list(z_bound))
        # same attribute names as the reference module (streamingflow.py:31-33, :41-43)
        self.bev_resolution = nn.Parameter(res, requires_grad=False)
        self.bev_start_position = nn.Parameter(start, requires_grad=False)
        self.bev_dimension = nn.Parameter(dim, requires_grad=False)
        self.final_dim, self.encoder_downsample, self.d_bound = tuple(final_dim), int(encoder_downsample), tuple(d_bound)
        self.frustum = self.create_frustum()
        self.depth_channels = self.frustum.shape[0]
        self.discount = float(discount)
        self._grid = (tuple(int(v) for v in dim.tolist()), tuple(float(v) for v in (start - res / 2.0).tolist()),
                      tuple(float(v) for v in res.tolist()))

    @classmethod
    def from_cfg(cfg_cls, cfg):
        """The keys ``streamingflow.__init__`` reads for this stage (streamingflow.py:28-43)."""
        return cfg_cls(cfg.LIFT.X_BOUND, cfg.LIFT.Y_BOUND, cfg.LIFT.Z_BOUND, cfg.LIFT.D_BOUND, cfg.IMAGE.FINAL_DIM,
                       cfg.MODEL.ENCODER.DOWNSAMPLE, cfg.LIFT.DISCOUNT)

    # ---- host-side geometry, as the reference computes it (tiny tensors) ------------------------
    def create_frustum(self):
        """streamingflow.py:149-168 -> Parameter [D, fH, fW, 3] = (pixel x, pixel y, depth)."""
        h, w = self.final_dim
        dh, dw = h // self.encoder_downsample, w // self.encoder_downsample
        depth_grid = torch.arange(*self.d_bound, dtype=torch.float).view(-1, 1, 1).expand(-1, dh, dw)
        n_d = depth_grid.shape[0]
        x_grid = torch.linspace(0, w - 1, dw, dtype=torch.float).view(1, 1, dw).expand(n_d, dh, dw)
        y_grid = torch.linspace(0, h - 1, dh, dtype=torch.float).view(1, dh, 1).expand(n_d, dh, dw)
        return nn.Parameter(torch.stack((x_grid, y_grid, depth_grid), -1), requires_grad=False)

    def get_geometry(self, intrinsics, extrinsics):
        """streamingflow.py:277-292 -> [B, N, D, fH, fW, 3] ego-frame positions (torch ops, as the reference)."""
        rotation, translation = extrinsics[..., :3, :3], extrinsics[..., :3, 3]
        B, N, _ = translation.shape
        points = self.frustum.unsqueeze(0).unsqueeze(0).unsqueeze(-1)
        points = torch.cat((points[:, :, :, :, :, :2] * points[:, :, :, :, :, 2:3], points[:, :, :, :, :, 2:3]), 5)
        combined = rotation.matmul(torch.inverse(intrinsics))
        points = combined.view(B, N, 1, 1, 1, 3, 3).matmul(points).squeeze(-1)
        points += translation.view(B, N, 1, 1, 1, 3)
        return points

    # ---- grid helpers ---------------------------------------------------------------------------
    def _grid_args(self):
        (X, Y, Z), lo, res = self._grid
        return (_lib.C.c_float * 3)(*lo), (_lib.C.c_float * 3)(*res), (_lib.C.c_int32 * 3)(X, Y, Z)

    def _index_geometry(self, geom, n_batch, want_coords=False):
        """geom [..., 3] fp32 cuda, points b-major -> (order, cell_start, coords|None)."""
        (X, Y, Z), _, _ = self._grid
        g = runtime.f32c(geom).view(-1, 3)
        n = g.shape[0]
        dev = g.device
        ncells = n_batch * X * Y * Z
        order = torch.empty((n,), dtype=torch.int32, device=dev)
        start = torch.empty((ncells + 1,), dtype=torch.int32, device=dev)
        coords = torch.empty((n, 4), dtype=torch.int32, device=dev) if want_coords else None
        L = _lib.lib()
        ws = runtime.workspace(L.sf_lift_index_ws_bytes(n, ncells), dev)
        lo, res, dim = self._grid_args()
        _lib.check(L.sf_lift_index_fwd(ptr(g), n, n_batch, lo, res, dim, ptr(coords), ptr(order), ptr(start), ptr(ws),
                                       ws.numel() * 4, runtime.stream_ptr(dev)), "lift_index")
        return order, start, coords

    # ---- reference API ------------------------------------------------------------------------------
    def bev_pool(self, geom_feats, x):
        """streamingflow.py:342-378.  geom_feats [B,N,D,H,W,3] float positions, x [B,N,D,H,W,C] ->
        (pooled [B, C, Z, X, Y], integer (x, y, z, b) of the points inside the grid, in point order).
        A frame with no point inside the grid gives zeros (the reference raises IndexError there)."""
        runtime.require_cuda(geom_feats, x)
        B, N, D, H, W, C = x.shape
        (X, Y, Z), _, _ = self._grid
        order, start, coords = self._index_geometry(geom_feats, B, want_coords=True)
        xf = runtime.f32c(x).view(-1, C)
        out = pool_cells(xf, order, start, B * Z * X * Y)
        kept = coords[coords[:, 0] >= 0].long()
        return out.view(B, Z, X, Y, C).permute(0, 4, 1, 2, 3).contiguous(), kept

    def warp_geometry(self, geometry_b, rotation_b, translation_b):
        """streamingflow.py:386-396 for one sample: frames 0..t are moved by pose t (t = 0..s-2),
        cumulatively, with the reference's own torch ops — on a copy (the reference updates the
        caller's tensor in place)."""
        geo = geometry_b.clone()
        s = geo.shape[0]
        for t in range(s - 1):
            tmp = rotation_b[t].view(1, 1, 1, 1, 1, 3, 3).matmul(geo[:t + 1].unsqueeze(-1)).squeeze(-1)
            tmp += translation_b[t].view(1, 1, 1, 1, 1, 3)
            geo[:t + 1] = tmp
        return geo

    def projection_to_birds_eye_view(self, x, geometry, future_egomotion, nhwc=False):
        """streamingflow.py:380-428.  x [b,s,n,d,h,w,c], geometry [b,s,n,d,h,w,3], future_egomotion [b,s,6]
        -> [b, s, c, X, Y] ([b, s, X, Y, c] with ``nhwc``)."""
        runtime.require_cuda(x, geometry, future_egomotion)
        batch, s, n, d, h, w, c = x.shape
        (X, Y, Z), _, _ = self._grid
        if Z != 1:
            raise RuntimeError("projection_to_birds_eye_view collapses the height axis: Z_BOUND must give one slice")
        mat = pose_vec2mat(future_egomotion)
        rotation, translation = mat[..., :3, :3], mat[..., :3, 3]
        out = torch.empty((batch, s, X * Y, c), dtype=torch.float32, device=x.device)
        xf = runtime.f32c(x)
        for b in range(batch):
            geo = self.warp_geometry(geometry[b], rotation[b], translation[b])
            order, start, _ = self._index_geometry(geo, s)          # the s frames of the sample in one sort
            for t in range(s):
                pool_cells(xf[b].view(-1, c), order, start[t * X * Y:], X * Y, prev=out[b, t - 1] if t else None,
                           discount=self.discount, out=out[b, t])
        out = out.view(batch, s, X, Y, c)
        return out if nhwc else runtime.to_nchw(out.view(batch * s, X, Y, c)).view(batch, s, c, X, Y)

    # ---- fused path -------------------------------------------------------------------------------
    def rig_affines(self, intrinsics, extrinsics, future_egomotion):
        """One 3x4 affine per (sample, frame, camera) taking (u*d, v*d, d, 1) to the final ego frame:
        R_cam K^-1 and the camera translation (get_geometry), then the cumulative ego-motion warps of
        projection_to_birds_eye_view, composed in float64 on the device.  intrinsics [b,s,n,3,3],
        extrinsics [b,s,n,4,4], future_egomotion [b,s,6] -> [b, s, n, 12] fp32."""
        b, s, n = intrinsics.shape[:3]
        R = extrinsics[..., :3, :3].double().matmul(torch.inverse(intrinsics.double()))
        A = torch.cat([R, extrinsics[..., :3, 3:].double()], -1)                 # [b,s,n,3,4]
        bottom = torch.zeros((b, s, n, 1, 4), dtype=torch.float64, device=A.device)
        bottom[..., 3] = 1.0
        A = torch.cat([A, bottom], -2)                                            # [b,s,n,4,4]
        ego = pose_vec2mat(future_egomotion.double())                             # [b,s,4,4]
        for t in range(s - 1):
            A[:, :t + 1] = ego[:, t].view(b, 1, 1, 4, 4).matmul(A[:, :t + 1])
        return A[..., :3, :].reshape(b, s, n, 12).float().contiguous()

    def lift_splat(self, feat, depth_logits, intrinsics, extrinsics, future_egomotion, nhwc=False):
        """get_geometry + depth softmax (x) features + projection_to_birds_eye_view in one pass.
        feat [b,s,n,C,fH,fW] image features, depth_logits [b,s,n,D,fH,fW] -> [b, s, C, X, Y]."""
        runtime.require_cuda(feat, depth_logits, intrinsics, extrinsics, future_egomotion)
        b, s, n, C, fH, fW = feat.shape
        D = depth_logits.shape[3]
        (X, Y, Z), _, _ = self._grid
        if Z != 1:
            raise RuntimeError("lift_splat collapses the height axis: Z_BOUND must give one slice")
        if (D, fH, fW) != tuple(self.frustum.shape[:3]):
            raise RuntimeError("feature / depth shape does not match the frustum %s" % (tuple(self.frustum.shape[:3]),))
        dev = feat.device
        L = _lib.lib()
        st = runtime.stream_ptr(dev)
        fHW = fH * fW
        rows = b * s * n
        logits = runtime.f32c(depth_logits).view(rows, D, fHW)
        prob = torch.empty_like(logits)
        _lib.check(L.sf_depth_softmax_fwd(ptr(logits), ptr(prob), rows, D, fHW, st), "depth_softmax")
        rays = runtime.to_nhwc(feat.reshape(rows, C, fH, fW)).view(rows * fHW, C)
        aff = self.rig_affines(intrinsics, extrinsics, future_egomotion).view(rows, 12)
        fr = self.frustum
        us, vs, ds = fr[0, 0, :, 0].contiguous(), fr[0, :, 0, 1].contiguous(), fr[:, 0, 0, 2].contiguous()
        npts = rows * D * fHW
        ncells = b * s * X * Y
        order = torch.empty((npts,), dtype=torch.int32, device=dev)
        start = torch.empty((ncells + 1,), dtype=torch.int32, device=dev)
        ws = runtime.workspace(L.sf_lift_index_ws_bytes(npts, ncells), dev)
        lo, res, dim = self._grid_args()
        _lib.check(L.sf_lift_index_rig_fwd(ptr(aff), ptr(us), ptr(vs), ptr(ds), b * s, n, D, fH, fW, lo, res, dim, ptr(order),
                                           ptr(start), ptr(ws), ws.numel() * 4, st), "lift_index_rig")
        out = torch.empty((b, s, X * Y, C), dtype=torch.float32, device=dev)
        for bi in range(b):
            for t in range(s):
                f = bi * s + t
                _lib.check(L.sf_lift_pool_fused_fwd(ptr(rays), ptr(prob), D, fHW, ptr(order), ptr(start[f * X * Y:]), X * Y, C,
                                                    ptr(out[bi, t - 1]) if t else ptr(None), self.discount, ptr(out[bi, t]), st),
                           "lift_pool_fused")
        out = out.view(b, s, X, Y, C)
        return out if nhwc else runtime.to_nchw(out.view(b * s, X, Y, C)).view(b, s, C, X, Y)
