"""CPU: the voxelisation oracle (oracle/voxelize.py) against fixtures produced by the reference's own
C++ CPU kernel + Voxelization module (tests/golden/voxelize.npz; `python -m oracle.gen_golden --only voxel`),
against its literal sequential formulation, and — when oracle/_ref holds the compiled reference — live."""
import numpy as np
import pytest
import torch

from util import cases, gold
from oracle import voxelize as VZ


@pytest.mark.parametrize("tag", list(cases.VOXEL_CASES))
def test_oracle_matches_reference_fixture(tag):
    G = gold("voxelize.npz")
    n, F, vs, rng, mp, mv = cases.VOXEL_CASES[tag]
    pts = cases.voxel_points(tag).numpy()
    v, c, k = VZ.hard_voxelize(pts, vs, rng, mp, mv)
    assert np.array_equal(v, G["voxels_" + tag]) and np.array_equal(c, G["coors_" + tag]) and np.array_equal(k, G["num_" + tag])
    v2, c2, k2 = VZ.hard_voxelize_loops(pts, vs, rng, mp, mv)
    assert np.array_equal(v, v2) and np.array_equal(c, c2) and np.array_equal(k, k2)


def test_oracle_matches_compiled_reference_live():
    from oracle import build_ref
    ext = build_ref.load_voxel_layer()
    if ext is None:
        pytest.skip("oracle/_ref/voxel_layer not built (needs /root/reference)")
    pts = torch.rand((4000, 5), generator=torch.Generator().manual_seed(9)) * 12 - 6
    vs, rng, mp, mv = [0.75, 0.75, 0.75], [-4.5, -4.5, -4.5, 4.5, 4.5, 4.5], 4, 300        # 12^3 cells
    voxels = pts.new_zeros((mv, mp, 5))
    coors = pts.new_zeros((mv, 3), dtype=torch.int)
    num = pts.new_zeros((mv,), dtype=torch.int)
    m = ext.hard_voxelize(pts, voxels, coors, num, vs, rng, mp, mv, 3, True)
    v, c, k = VZ.hard_voxelize(pts.numpy(), vs, rng, mp, mv)
    assert m == v.shape[0]
    assert np.array_equal(voxels[:m].numpy(), v) and np.array_equal(coors[:m].numpy(), c) and np.array_equal(num[:m].numpy(), k)


def test_shipped_grid_properties():
    """1600 x 1600 x 40 grid (the reference's CPU kernel is not memory-safe there): restatement only."""
    vs, rng, mp, mv = cases.VOXEL_SHIPPED
    assert VZ.grid_size(vs, rng) == [1600, 1600, 40]      # the sparse encoder pads z to 41
    g = torch.Generator().manual_seed(2)
    pts = torch.cat([torch.randn((20000, 3), generator=g) * torch.tensor([12.0, 12.0, 1.5]), torch.rand((20000, 2), generator=g)], 1)
    pts = torch.cat([pts, pts[:5000] + 1e-4], 0).numpy()                    # near-duplicates share voxels
    v, c, k = VZ.hard_voxelize(pts, vs, rng, mp, mv)
    _, ok = VZ.point_coors(pts, vs, rng)
    assert int(k.sum()) <= int(ok.sum()) and (k >= 1).all() and (k <= mp).all()
    assert len({tuple(r) for r in c.tolist()}) == c.shape[0]               # one voxel per coordinate
    first = v[:, 0, :]                                                      # first point of every voxel, in appearance order
    idx = [int(np.nonzero((pts == f).all(1))[0][0]) for f in first[:200]]
    assert idx == sorted(idx)
    feats, coords, sizes = VZ.sf_voxelize([torch.from_numpy(pts), torch.from_numpy(pts[:1000])], vs, rng, mp, mv)
    assert feats.shape[0] == coords.shape[0] == sizes.shape[0] and set(coords[:, 0].tolist()) == {0, 1}


@pytest.mark.parametrize("tag", list(cases.VOXEL_CASES))
def test_dynamic_voxelize_oracle_matches_reference_fixture(tag):
    """the max_points == -1 branch (voxelize.py:46-49): per-point coordinates from the reference module on its own C++ kernel"""
    G = gold("voxelize.npz")
    n, F, vs, rng, mp, mv = cases.VOXEL_CASES[tag]
    c = VZ.dynamic_voxelize(cases.voxel_points(tag).numpy(), vs, rng)
    assert c.dtype == np.int32 and np.array_equal(c, G["dyn_coors_" + tag])
    assert ((c == -1).all(1) | (c >= 0).all(1)).all()          # a point is inside on every axis or marked on every axis


def test_dynamic_voxelize_oracle_matches_compiled_reference_live():
    from oracle import build_ref
    ext = build_ref.load_voxel_layer()
    if ext is None:
        pytest.skip("oracle/_ref/voxel_layer not built (needs /root/reference)")
    pts = torch.rand((4000, 5), generator=torch.Generator().manual_seed(9)) * 12 - 6
    vs, rng = [0.75, 0.75, 0.75], [-4.5, -4.5, -4.5, 4.5, 4.5, 4.5]
    coors = pts.new_zeros((pts.shape[0], 3), dtype=torch.int)
    ext.dynamic_voxelize(pts, coors, vs, rng, 3)
    assert np.array_equal(coors.numpy(), VZ.dynamic_voxelize(pts.numpy(), vs, rng))
