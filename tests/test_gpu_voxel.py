"""GPU: LiDAR hard voxelisation through libsfnative vs the oracle and vs fixtures from the reference's own
C++ kernel.  Integer / copy work: everything must be bit-exact (the per-voxel mean is fp32: 1e-6)."""
import numpy as np
import pytest
import torch

from util import cases, gold, maxabs
from oracle import voxelize as VZ

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tag", list(cases.VOXEL_CASES))
def test_voxelization_module(tag):
    from streamingflow_amd.voxelize import Voxelization
    G = gold("voxelize.npz")
    n, F, vs, rng, mp, mv = cases.VOXEL_CASES[tag]
    pts = cases.voxel_points(tag)
    m = Voxelization(list(vs), list(rng), mp, (mv, mv)).eval()
    v, c, k = m(pts.cuda())
    assert c.dtype == torch.int32 and k.dtype == torch.int32
    assert np.array_equal(v.cpu().numpy(), G["voxels_" + tag])
    assert np.array_equal(c.cpu().numpy(), G["coors_" + tag])
    assert np.array_equal(k.cpu().numpy(), G["num_" + tag])
    with pytest.raises(RuntimeError):
        m(pts)                                           # CPU tensor: no fallback


def test_training_uses_first_cap():
    from streamingflow_amd.voxelize import Voxelization
    n, F, vs, rng, mp, mv = cases.VOXEL_CASES["cube16"]
    pts = cases.voxel_points("cube16")
    m = Voxelization(list(vs), list(rng), mp, (20, 200))
    m.train()
    v, c, k = m(pts.cuda())
    w, d, q = VZ.hard_voxelize(pts.numpy(), vs, rng, mp, 20)
    assert np.array_equal(v.cpu().numpy(), w) and np.array_equal(c.cpu().numpy(), d) and np.array_equal(k.cpu().numpy(), q)


def test_streamingflow_voxelize_mean():
    from streamingflow_amd.voxelize import Voxelization, voxelize
    n, F, vs, rng, mp, mv = cases.VOXEL_CASES["cube8_dense"]
    clouds = [cases.voxel_points("cube8_dense"), cases.voxel_points("cube8_dense")[:700] * 0.5, cases.voxel_points("cube8_dense")[:1] + 100.0]
    m = Voxelization(list(vs), list(rng), mp, (mv, mv)).eval()
    feats, coords, sizes = voxelize([p.cuda() for p in clouds], m)
    wf, wc, wsz = VZ.sf_voxelize([p.numpy() for p in clouds], vs, rng, mp, mv)
    assert torch.equal(coords.cpu(), wc) and torch.equal(sizes.cpu(), wsz)
    assert maxabs(feats, wf) <= 1e-6
    # un-reduced variant returns the padded voxel tensor rows
    f2, c2, s2 = voxelize([p.cuda() for p in clouds[:2]], m, voxelize_reduce=False)
    assert f2.dim() == 3 and torch.equal(c2.cpu(), wc[: c2.shape[0]])


def test_shipped_size_exact_and_reproducible():
    """350 000 x 5 points, 1600 x 1600 x 40 grid, <= 10 points / voxel, <= 160 000 voxels."""
    from streamingflow_amd.voxelize import Voxelization
    vs, rng, mp, mv = cases.VOXEL_SHIPPED
    g = torch.Generator().manual_seed(7)
    n = 350000
    pts = torch.cat([torch.randn((n, 3), generator=g) * torch.tensor([15.0, 15.0, 1.2]), torch.rand((n, 2), generator=g)], 1)
    pts[n // 2:] = pts[: n - n // 2] + torch.randn((n - n // 2, 5), generator=g) * 0.02       # clustered: shared voxels
    pts[-20000:] = 0.0                                                                         # padded rows (NuscenesData.py:869-873)
    m = Voxelization(list(vs), list(rng), mp, (120000, mv)).eval()
    v, c, k = m(pts.cuda())
    v2, c2, k2 = m(pts.cuda())
    assert torch.equal(v, v2) and torch.equal(c, c2) and torch.equal(k, k2)
    w, d, q = VZ.hard_voxelize(pts.numpy(), vs, rng, mp, mv)
    assert v.shape[0] == w.shape[0]
    assert np.array_equal(c.cpu().numpy(), d) and np.array_equal(k.cpu().numpy(), q) and np.array_equal(v.cpu().numpy(), w)


@pytest.mark.parametrize("tag", list(cases.VOXEL_CASES))
def test_dynamic_voxelization_module(tag):
    """max_num_points == -1 (voxelize.py:46-49): the module returns the per-point coordinates only — bit-exact against the fixture from
    the reference module on its own C++ kernel; also at the shipped grid against the oracle, with points outside the range on every axis."""
    from streamingflow_amd.voxelize import Voxelization, dynamic_voxelize
    G = gold("voxelize.npz")
    n, F, vs, rng, mp, mv = cases.VOXEL_CASES[tag]
    pts = cases.voxel_points(tag)
    c = Voxelization(list(vs), list(rng), -1, (mv, mv)).eval()(pts.cuda())
    assert c.dtype == torch.int32 and tuple(c.shape) == (pts.shape[0], 3)
    assert np.array_equal(c.cpu().numpy(), G["dyn_coors_" + tag])
    with pytest.raises(RuntimeError):
        dynamic_voxelize(pts, vs, rng)                   # CPU tensor: no fallback
    if tag == "cube16":
        vs2, rng2, _, _ = cases.VOXEL_SHIPPED
        g = torch.Generator().manual_seed(5)
        big = torch.cat([torch.randn((350000, 3), generator=g) * torch.tensor([30.0, 30.0, 3.0]), torch.rand((350000, 2), generator=g)], 1)
        got = dynamic_voxelize(big.cuda(), vs2, rng2).cpu().numpy()
        want = VZ.dynamic_voxelize(big.numpy(), vs2, rng2)
        assert np.array_equal(got, want) and (want == -1).any() and (want >= 0).any()
