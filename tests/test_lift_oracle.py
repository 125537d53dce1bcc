"""CPU: the lift-splat oracle (oracle/lift_splat.py) against the fixtures the REAL reference Python
produced (tests/golden/lift_splat.npz, made by `python -m oracle.gen_golden --only lift`), and the
host-side geometry helpers of the product against the oracle."""
import numpy as np
import pytest
import torch

from util import cases, gold, hashfill, maxabs
from oracle import lift_splat as LS


@pytest.fixture(scope="module")
def G():
    return gold("lift_splat.npz")


@pytest.mark.parametrize("tag", [t for t in cases.LIFT_POOL_CASES if t != "empty"])
def test_sf_bev_pool_matches_reference(G, tag):
    geo, x, start, res, dim = cases.lift_pool_inputs(tag)
    out, kept = LS.sf_bev_pool(geo, x, start, res, dim, stable=False)
    assert np.array_equal(kept.numpy().astype(np.int32), G["pool_kept_" + tag])         # integer work: exact
    assert maxabs(out, G["pool_" + tag]) <= 1e-5
    out_s, kept_s = LS.sf_bev_pool(geo, x, start, res, dim, stable=True)                  # the HIP path's sum order
    assert torch.equal(kept, kept_s)
    assert maxabs(out_s, G["pool_" + tag]) <= 1e-5


@pytest.mark.parametrize("tag", list(cases.LIFT_CASES))
def test_projection_matches_reference(G, tag):
    feat, depth, geo, ego, (start, res, dim), discount = cases.lift_inputs(tag)
    b, s, n, C, fH, fW = feat.shape
    x = LS.depth_outer(feat.reshape(b * s * n, C, fH, fW), depth.reshape(b * s * n, -1, fH, fW))
    x = x.reshape(b, s, n, *x.shape[1:])
    geo0 = geo.clone()
    out = LS.projection_to_birds_eye_view(x, geo, ego, start, res, dim, discount, stable=True)
    assert torch.equal(geo, geo0)                      # the oracle does not touch the caller's geometry
    assert maxabs(out, G["proj_" + tag]) <= 1e-5


def test_frustum_geometry_pose(G):
    fr = LS.create_frustum((32, 48), 8, [2.0, 10.0, 1.0])
    assert np.array_equal(fr.numpy(), G["frustum"])
    intr = torch.tensor([[20.0, 0.0, 24.0], [0.0, 20.0, 16.0], [0.0, 0.0, 1.0]]).repeat(1, 2, 1, 1)
    ang = hashfill.uniform("lift_extr_r", (1, 2, 3), -1.0, 1.0, seed=31)
    extr = LS.pose_vec2mat(torch.cat([hashfill.uniform("lift_extr_t", (1, 2, 3), -1.0, 1.0, seed=32), ang], -1))
    assert maxabs(extr, G["pose_vec2mat"]) <= 1e-7
    assert maxabs(LS.get_geometry(fr, intr, extr), G["geometry"]) <= 1e-5


def test_pool_op_and_kernel_restatement(G):
    n, c = 3000, 8
    coords = (hashfill.uniform("lift_op_coords", (n, 4), 0.0, 1.0, seed=33) * torch.tensor([7.0, 6.0, 2.0, 2.0])).long()
    feats = hashfill.normal("lift_op_feats", (n, c), seed=34)
    a = LS.bev_pool_op(feats, coords, 2, 2, 7, 6, stable=True)
    assert maxabs(a, G["op_bev_pool"]) <= 1e-5
    assert maxabs(a, G["op_quickcumsum"]) <= 1e-4          # the reference's cumsum implementation: looser by construction
    # vectorised kernel == the literal loops (sequential fp32 adds), bit for bit
    ranks = coords[:, 0] * (6 * 2 * 2) + coords[:, 1] * (2 * 2) + coords[:, 2] * 2 + coords[:, 3]
    idx = torch.argsort(ranks, stable=True)
    f2, c2, r2 = feats[idx], coords[idx], ranks[idx]
    kept = torch.ones(n, dtype=torch.bool)
    kept[1:] = r2[1:] != r2[:-1]
    st = torch.where(kept)[0].int()
    ln = torch.zeros_like(st)
    ln[:-1] = st[1:] - st[:-1]
    ln[-1] = n - st[-1]
    assert torch.equal(LS.bev_pool_kernel_loops(f2, c2.int(), ln, st, 2, 2, 7, 6), LS.bev_pool_kernel(f2, c2.int(), ln, st, 2, 2, 7, 6))


def test_quantise_truncates_toward_zero():
    res, start, dim = LS.calculate_birds_eye_view_parameters([-4.0, 4.0, 0.5], [-4.0, 4.0, 0.5], [-10.0, 10.0, 20.0])
    g = torch.tensor([[-4.2, -4.0, 0.0], [-4.6, 3.99, 9.9], [4.0, 0.0, -10.1], [-3.75, 0.26, 0.0]])
    q = LS.quantise(g, start, res)
    # -4.2 -> (-0.4).long() == 0 (kept, as in the reference); -4.6 -> -1 (dropped); 4.0 -> 16 (dropped)
    assert q.tolist() == [[0, 0, 0], [-1, 15, 0], [16, 8, 0], [0, 8, 0]]


def test_product_host_geometry_matches_oracle():
    """LiftSplat's torch-side helpers (frustum, get_geometry, pose_vec2mat, composed rig affines) on CPU."""
    from streamingflow_amd.models.lift_splat import LiftSplat, pose_vec2mat
    tag = "rig_small"
    feat, depth, intr, extr, ego, fr, (start, res, dim), discount = cases.lift_rig_inputs(tag)
    b, s, n, C, final_dim, down, d_bound, xb, yb, zb, _ = cases.LIFT_RIG_CASES[tag]
    m = LiftSplat(xb, yb, zb, d_bound, final_dim, down, discount)
    assert torch.equal(m.frustum.data, fr)
    assert torch.equal(m.bev_dimension.data, dim) and torch.equal(m.bev_start_position.data, start)
    assert torch.equal(pose_vec2mat(ego), LS.pose_vec2mat(ego))
    g = LS.get_geometry(fr, intr.view(b * s, n, 3, 3), extr.view(b * s, n, 4, 4))
    assert torch.equal(m.get_geometry(intr.view(b * s, n, 3, 3), extr.view(b * s, n, 4, 4)), g)
    # composed affine == sequential reference chain (float64 evaluation of both)
    A = m.rig_affines(intr, extr, ego).double().view(b, s, n, 3, 4)
    g64 = LS.get_geometry(fr.double(), intr.double().view(b * s, n, 3, 3), extr.double().view(b * s, n, 4, 4)).view(b, s, n, *fr.shape)
    mat = LS.pose_vec2mat(ego.double())
    fin = torch.stack([LS.warp_geometry(g64[i], mat[i, :, :3, :3], mat[i, :, :3, 3]) for i in range(b)])
    pts = torch.cat((fr[..., :2] * fr[..., 2:3], fr[..., 2:3], torch.ones_like(fr[..., :1])), -1).double()     # [D,fH,fW,4]
    mine = torch.einsum("bsnij,dhwj->bsndhwi", A, pts)
    assert float((mine - fin).abs().max()) <= 1e-5
    with pytest.raises(RuntimeError):
        m.bev_pool(torch.zeros(1, 1, 1, 1, 1, 3), torch.zeros(1, 1, 1, 1, 1, 8))      # CPU tensors: no fallback
