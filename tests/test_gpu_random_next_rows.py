"""GPU: randomised sweeps of the integer/gather kernels of the "next" rows against their oracles (fixed seeds):
bev_pool with odd channel counts / several batches / height slices, hard voxelisation with random caps and grids,
sparse neighbour tables and output sites for random kernel / stride / padding, confusion matrices."""
import random

import numpy as np
import pytest
import torch

from oracle import lift_splat as LS, sparse_encoder_ref as SR, voxelize as VZ

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("i", range(12))
def test_random_bev_pool(i):
    from streamingflow_amd.bev_pool import bev_pool
    r = random.Random(200 + i)
    B, D, H, W = r.choice([1, 2, 3]), r.choice([1, 2, 5]), r.choice([3, 16, 40]), r.choice([4, 17, 33])
    C = r.choice([1, 3, 8, 64, 65, 130])
    n = r.choice([1, 50, 4000, 30000])
    g = torch.Generator().manual_seed(i)
    coords = torch.stack([torch.randint(0, H, (n,), generator=g), torch.randint(0, W, (n,), generator=g),
                          torch.randint(0, D, (n,), generator=g), torch.randint(0, B, (n,), generator=g)], 1)
    if i % 3 == 0:          # crowd a few cells
        coords[: n // 2] = coords[0]
    feats = torch.randn((n, C), generator=g)
    out = bev_pool(feats.cuda(), coords.cuda(), B, D, H, W)
    assert torch.equal(out.cpu(), LS.bev_pool_op(feats, coords, B, D, H, W, stable=True))


@pytest.mark.parametrize("i", range(10))
def test_random_voxelize(i):
    from streamingflow_amd.voxelize import Voxelization
    r = random.Random(300 + i)
    F = r.choice([3, 4, 5, 7])
    vs = [r.choice([0.25, 0.5, 1.0]), r.choice([0.25, 0.5]), r.choice([0.2, 0.5, 2.0])]
    rng = [-r.choice([3.0, 8.0]), -r.choice([2.0, 8.0]), -r.choice([1.0, 5.0]), r.choice([3.0, 8.0]), r.choice([2.0, 8.0]), r.choice([1.0, 3.0])]
    mp, mv = r.choice([1, 3, 10]), r.choice([5, 100, 5000])
    n = r.choice([1, 17, 3000, 60000])
    pts = (torch.rand((n, F), generator=torch.Generator().manual_seed(i)) - 0.5) * 20.0
    if i % 2:
        pts[n // 3:] = pts[: n - n // 3] * 0.25                  # dense clusters: per-voxel cap is hit
    m = Voxelization(vs, rng, mp, (mv, mv)).eval()
    v, c, k = m(pts.cuda())
    w, d, q = VZ.hard_voxelize(pts.numpy(), vs, rng, mp, mv)
    assert np.array_equal(v.cpu().numpy(), w) and np.array_equal(c.cpu().numpy(), d) and np.array_equal(k.cpu().numpy(), q)


@pytest.mark.parametrize("i", range(10))
def test_random_sparse_index(i):
    from streamingflow_amd.models.sparse_encoder import SparseEncoder
    r = random.Random(400 + i)
    shape = [r.choice([8, 21, 40]), r.choice([9, 16, 33]), r.choice([5, 11, 41])]
    B = r.choice([1, 2, 3])
    k = [r.choice([1, 3]), r.choice([1, 3]), r.choice([1, 3])]
    s = [r.choice([1, 2]), r.choice([1, 2]), r.choice([1, 2])]
    p = [r.choice([0, 1]) if k[a] == 3 else 0 for a in range(3)]
    n = r.choice([1, 40, 900])
    g = torch.Generator().manual_seed(i)
    cells = torch.randperm(B * shape[0] * shape[1] * shape[2], generator=g)[:n]
    z = cells % shape[2]; y = (cells // shape[2]) % shape[1]; x = (cells // (shape[2] * shape[1])) % shape[0]; b = cells // (shape[0] * shape[1] * shape[2])
    coords = torch.stack([b, x, y, z], 1).int()
    enc = SparseEncoder.__new__(SparseEncoder)                 # only the index helpers are exercised
    cd = coords.cuda()
    oc, so = SparseEncoder._out_sites(enc, cd, B, shape, k, s, p)
    wc, wso = SR.down_sites(coords.numpy(), shape, k, s, p)
    assert so == wso and np.array_equal(oc.cpu().numpy(), wc)
    tab = SparseEncoder._table(enc, cd, oc.contiguous(), B, shape, k, s, p, False)
    want = SR.neighbour_table(coords.numpy(), shape, wc, k, s, p, False)
    assert np.array_equal(tab.cpu().numpy()[: wc.shape[0]], want.astype(np.int32))
    tabs = SparseEncoder._table(enc, cd, cd, B, shape, [3, 3, 3], [1, 1, 1], [0, 0, 0], True)
    assert np.array_equal(tabs.cpu().numpy(), SR.neighbour_table(coords.numpy(), shape, coords.numpy(), [3, 3, 3], [1, 1, 1], [0, 0, 0], True).astype(np.int32))


@pytest.mark.parametrize("i", range(6))
def test_random_confusion(i):
    from streamingflow_amd.metrics import confusion
    r = random.Random(500 + i)
    K, n = r.choice([2, 3, 17, 120]), r.choice([1, 1000, 300000])
    g = torch.Generator().manual_seed(i)
    a, b = torch.randint(0, K, (n,), generator=g), torch.randint(0, K, (n,), generator=g)
    conf, bad = confusion(a.cuda(), b.cuda(), K)
    want = torch.bincount(a + K * b, minlength=K * K).view(K, K)
    assert int(bad.item()) == 0 and torch.equal(conf.cpu(), want)
