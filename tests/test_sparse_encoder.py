"""LiDAR SparseEncoder (second half).  The compiled spconv of the reference cannot be built
here, so the oracle is a restatement of spconv 1.x's semantics ("parity unpinned"): two independent
formulations (sparse lookup vs dense conv3d + masks) are checked against each other on the CPU, the HIP path
against the sparse one on the GPU."""
import numpy as np
import pytest
import torch

from util import cases, maxabs
from oracle import sparse_encoder_ref as SR


@pytest.mark.parametrize("tag", list(cases.SPARSE_CASES))
def test_sparse_and_dense_formulations_agree(tag):
    cfg = cases.sparse_cfg(tag)
    f, c, B = cases.sparse_inputs(tag)
    sd = cases.sparse_state_dict(tag)
    a = SR.sparse_encoder_forward(sd, f.numpy(), c.numpy(), B, cfg)
    b = SR.sparse_encoder_forward_dense(sd, f.numpy(), c.numpy(), B, cfg)
    assert a.shape == b.shape and float(a.abs().max()) > 0.05
    assert maxabs(a, b) <= 1e-5


def test_output_sites_of_a_strided_conv():
    coords = np.array([[0, 0, 0, 0], [0, 3, 2, 1], [1, 7, 7, 7]], np.int32)
    out, so = SR.down_sites(coords, [8, 8, 8], [3, 3, 3], [2, 2, 2], [1, 1, 1])
    assert so == [4, 4, 4]
    # (3,2,1): o = (p + 1 - k)/2 for k in 0..2 where divisible -> x: 2 (k=0), 1 (k=2); y: 1 (k=1); z: 1 (k=0), 0 (k=2)
    want = {(0, 0, 0, 0), (0, 2, 1, 1), (0, 2, 1, 0), (0, 1, 1, 1), (0, 1, 1, 0), (1, 3, 3, 3)}      # (7+1-k)/2: k=0 -> 4 (outside), k=2 -> 3
    assert {tuple(r) for r in out.tolist()} == want


def test_product_state_dict_names():
    from streamingflow_amd.models.sparse_encoder import SparseEncoder
    cfg = SR.default_cfg()
    m = SparseEncoder(cfg["in_channels"], cfg["sparse_shape"], base_channels=cfg["base_channels"], output_channels=cfg["output_channels"],
                      encoder_channels=cfg["encoder_channels"], encoder_paddings=cfg["encoder_paddings"], block_type="basicblock")
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == SR.state_dict_shapes(cfg)
    with pytest.raises(AssertionError):
        SparseEncoder(5, [8, 8, 8], block_type="bottleneck")


def _build(tag):
    from streamingflow_amd.models.sparse_encoder import SparseEncoder
    cfg = cases.sparse_cfg(tag)
    m = SparseEncoder(cfg["in_channels"], cfg["sparse_shape"], base_channels=cfg["base_channels"], output_channels=cfg["output_channels"],
                      encoder_channels=cfg["encoder_channels"], encoder_paddings=cfg["encoder_paddings"],
                      block_type=cfg.get("block_type", "basicblock")).eval()
    sd = cases.sparse_state_dict(tag)
    m.load_state_dict(sd)
    return m.cuda(), sd, cfg


@pytest.mark.gpu
@pytest.mark.parametrize("sort", [False, True])
@pytest.mark.parametrize("tag", list(cases.SPARSE_CASES))
def test_gpu_forward(tag, sort, monkeypatch):
    """sort=True: the opt-in form with the rows of every stage in neighbour-mask order and per-tile tap masks (sf_sparse_conv_masked_fwd):
    a dropped tap would have gathered zero rows, so the output must equal the stored-order output BIT FOR BIT."""
    from streamingflow_amd.models.sparse_encoder import SparseEncoder
    m, sd, cfg = _build(tag)
    f, c, B = cases.sparse_inputs(tag)
    if sort:
        plain = m(f.cuda(), c.cuda(), B)
        monkeypatch.setattr(SparseEncoder, "SORT", True)
        assert torch.equal(m(f.cuda(), c.cuda(), B), plain)
    out = m(f.cuda(), c.cuda(), B)
    want = SR.sparse_encoder_forward(sd, f.numpy(), c.numpy(), B, cfg)
    err = maxabs(out, want)
    print(tag, tuple(out.shape), "max-abs", err)
    assert err <= 1e-4
    # site order must not matter (dense() removes it): a permutation of the input rows gives the same grid
    perm = torch.randperm(f.shape[0], generator=torch.Generator().manual_seed(3))
    out2 = m(f[perm].cuda(), c[perm].cuda(), B)
    assert maxabs(out2, out) <= 1e-5
    nh = m(f.cuda(), c.cuda(), B, nhwc=True)
    assert torch.equal(nh.permute(0, 3, 1, 2).contiguous(), out)


@pytest.mark.gpu
def test_gpu_index_kernels_exact():
    """Neighbour tables and output sites are integer work: exact vs the oracle."""
    from streamingflow_amd.models.sparse_encoder import SparseEncoder
    tag = "grid32x24x27"
    m, _, cfg = _build(tag)
    f, c, B = cases.sparse_inputs(tag)
    shape = cfg["sparse_shape"]
    cd = c.cuda()
    tab = m._table(cd, cd, B, shape, [3, 3, 3], [1, 1, 1], [0, 0, 0], True)
    want = SR.neighbour_table(c.numpy(), shape, c.numpy(), [3, 3, 3], [1, 1, 1], [0, 0, 0], True)
    assert np.array_equal(tab.cpu().numpy(), want.astype(np.int32))
    oc, so = m._out_sites(cd, B, shape, [3, 3, 3], [2, 2, 2], [1, 1, 1])
    wc, wso = SR.down_sites(c.numpy(), shape, [3, 3, 3], [2, 2, 2], [1, 1, 1])
    assert so == wso and np.array_equal(oc.cpu().numpy(), wc)
    tab2 = m._table(cd, oc, B, shape, [3, 3, 3], [2, 2, 2], [1, 1, 1], False)
    want2 = SR.neighbour_table(c.numpy(), shape, wc, [3, 3, 3], [2, 2, 2], [1, 1, 1], False)
    assert np.array_equal(tab2.cpu().numpy(), want2.astype(np.int32))


@pytest.mark.gpu
def test_gpu_shipped_size_properties():
    """The shipped LiDAR configuration (1600 x 1600 x 41 cells, 350 000-point cloud -> 160 000-voxel cap, the reference's
    channel widths): what can be checked without a buildable spconv.  (1) the index kernels are exact against the numpy
    restatement at this size: neighbour table of the first submanifold layer and the output sites / table of the first
    strided layer; (2) the forward is invariant under a permutation of the input rows; (3) every output value is finite
    and the occupied BEV cells are exactly the cells under an active site of the last layer."""
    import os
    import sys
    from util import ROOT
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import voxelbench
    from streamingflow_amd.models.sparse_encoder import SparseEncoder
    from streamingflow_amd.voxelize import Voxelization, voxelize
    cfg = SR.default_cfg()
    m = SparseEncoder(cfg["in_channels"], cfg["sparse_shape"], base_channels=cfg["base_channels"], output_channels=cfg["output_channels"],
                      encoder_channels=cfg["encoder_channels"], encoder_paddings=cfg["encoder_paddings"], block_type="basicblock").eval()
    from util import hashfill
    m.load_state_dict(hashfill.fill_state_dict(m.state_dict(), seed=83, gain=1.6))
    m = m.cuda()
    vox = Voxelization(voxel_size=[0.0625, 0.0625, 0.2], point_cloud_range=[-50.0, -50.0, -5.0, 50.0, 50.0, 3.0], max_num_points=10,
                       max_voxels=(160000, 160000))
    pts = voxelbench.cloud(350000, seed=7).cuda()
    feats, coords, _ = voxelize([pts], vox, True)
    n = coords.shape[0]
    assert 100000 <= n <= 160000 and feats.shape == (n, 5)
    shape = cfg["sparse_shape"]
    cn = coords.cpu().numpy()
    # (1) integer work, exact
    tab = m._table(coords.int().contiguous(), coords.int().contiguous(), 1, shape, [3, 3, 3], [1, 1, 1], [0, 0, 0], True)
    assert np.array_equal(tab.cpu().numpy(), SR.neighbour_table(cn, shape, cn, [3, 3, 3], [1, 1, 1], [0, 0, 0], True).astype(np.int32))
    oc, so = m._out_sites(coords.int().contiguous(), 1, shape, [3, 3, 3], [2, 2, 2], [1, 1, 1])
    wc, wso = SR.down_sites(cn, shape, [3, 3, 3], [2, 2, 2], [1, 1, 1])
    assert so == wso and np.array_equal(oc.cpu().numpy(), wc)
    tab2 = m._table(coords.int().contiguous(), oc, 1, shape, [3, 3, 3], [2, 2, 2], [1, 1, 1], False)
    assert np.array_equal(tab2.cpu().numpy(), SR.neighbour_table(cn, shape, wc, [3, 3, 3], [2, 2, 2], [1, 1, 1], False).astype(np.int32))
    # (2) permutation invariance of the whole forward
    out = m(feats, coords, 1)
    assert out.shape == (1, 256, 200, 200) and torch.isfinite(out).all()
    perm = torch.randperm(n, generator=torch.Generator().manual_seed(11)).cuda()
    out2 = m(feats[perm], coords[perm], 1)
    assert maxabs(out2, out) <= 2e-5
    # (3) occupancy: a BEV cell is non-zero only under an active site of the last layer (site sets are exact integer work)
    sites = cn
    sh = list(shape)
    for k, s, p in (([3, 3, 3], [2, 2, 2], [1, 1, 1]), ([3, 3, 3], [2, 2, 2], [1, 1, 1]), ([3, 3, 3], [2, 2, 2], [1, 1, 0]), ([1, 1, 3], [1, 1, 2], [0, 0, 0])):
        sites, sh = SR.down_sites(sites, sh, k, s, p)
    occ = np.zeros((200, 200), bool)
    occ[sites[:, 1], sites[:, 2]] = True
    nz = (out[0].abs().sum(0) > 0).cpu().numpy()
    assert not (nz & ~occ).any() and nz.sum() >= 0.95 * occ.sum()
