"""TEST INFRASTRUCTURE — CPU restatement of the LiDAR ``SparseEncoder`` (second half).

Not imported by the product.  Follows mmdet3d/models/backbones/sparse_encoder.py:36-139 (+ :141-226
``make_encoder_layers``), mmdet3d/ops/sparse_block.py:61-107 (``SparseBasicBlock``), :110-176
(``make_sparse_convmodule``) and the Python layer of the vendored spconv 1.x
(mmdet3d/ops/spconv/conv.py:114-214, ops.py:19-33 output size, structure.py ``dense``).

**Parity unpinned.**  The arithmetic lives in the compiled extension ``sparse_conv_ext``
(mmdet3d/ops/spconv/src, CUDA + cmake-era C++ with mmcv/pybind glue) which cannot be built here, and
the Python above it needs mmcv / mmdet (absent).  This file restates spconv 1.x's published semantics:
  SubMConv3d      output sites = input sites; out[p] = sum_k W[k] . in[p + k - k//2]        (padding ignored)
  SparseConv3d    out shape by the dense formula; output sites = every o with 0 <= o < out_shape for which
                  some active input p and kernel offset k satisfy  p = o*stride - padding + k  (dilation 1);
                  out[o] = sum_k W[k] . in[o*stride - padding + k]
  weight layout   [kx][ky][kz][Cin][Cout], no bias; the site order of the outputs is an implementation
                  detail that ``dense()`` removes.
Two independent formulations live here and are checked against each other (tests/test_sparse_oracle.py):
a sparse one (sorted keys + searchsorted, any grid size) and a dense one (``torch.conv3d`` on the
densified grid + activity masks, small grids only).
"""
import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-3      # norm_cfg=dict(type="BN1d", eps=1e-3, momentum=0.01), sparse_encoder.py:40


def _triple(v):
    return [int(x) for x in v] if isinstance(v, (list, tuple)) else [int(v)] * 3


def out_shape(shape, k, s, p):
    """spconv/ops.py:19-33 (dilation 1)."""
    return [(shape[i] + 2 * p[i] - (k[i] - 1) - 1) // s[i] + 1 for i in range(3)]


def _keys(coords, shape):
    c = coords.astype(np.int64)
    return ((c[:, 0] * shape[0] + c[:, 1]) * shape[1] + c[:, 2]) * shape[2] + c[:, 3]


def _lookup(sorted_keys, perm, q, valid):
    pos = np.searchsorted(sorted_keys, q)
    pos = np.minimum(pos, len(sorted_keys) - 1)
    hit = valid & (sorted_keys[pos] == q)
    return np.where(hit, perm[pos], -1)


def neighbour_table(coords_in, shape_in, coords_out, k, s, p, subm):
    """[n_out, ntaps] index of the input site feeding output site j through kernel offset t (or -1);
    t = (kx*KY + ky)*KZ + kz."""
    keys = _keys(coords_in, shape_in)
    perm = np.argsort(keys, kind="stable")
    sk = keys[perm]
    n_out = coords_out.shape[0]
    tab = np.full((n_out, k[0] * k[1] * k[2]), -1, dtype=np.int64)
    co = coords_out.astype(np.int64)
    t = 0
    for kx in range(k[0]):
        for ky in range(k[1]):
            for kz in range(k[2]):
                off = np.array([kx, ky, kz])
                if subm:
                    q = co[:, 1:] + off - np.array([k[0] // 2, k[1] // 2, k[2] // 2])
                else:
                    q = co[:, 1:] * np.array(s) - np.array(p) + off
                valid = ((q >= 0) & (q < np.array(shape_in))).all(1)
                qk = ((co[:, 0] * shape_in[0] + q[:, 0]) * shape_in[1] + q[:, 1]) * shape_in[2] + q[:, 2]
                tab[:, t] = _lookup(sk, perm, np.where(valid, qk, 0), valid)
                t += 1
    return tab


def down_sites(coords_in, shape_in, k, s, p):
    """Output sites of a SparseConv3d, sorted by (b, x, y, z)."""
    so = out_shape(shape_in, k, s, p)
    c = coords_in.astype(np.int64)
    cand = []
    for kx in range(k[0]):
        for ky in range(k[1]):
            for kz in range(k[2]):
                num = c[:, 1:] + np.array(p) - np.array([kx, ky, kz])
                ok = (num % np.array(s) == 0).all(1)
                o = num // np.array(s)
                ok &= ((o >= 0) & (o < np.array(so))).all(1)
                cand.append(np.concatenate([c[ok, :1], o[ok]], 1))
    cand = np.concatenate(cand, 0)
    if cand.shape[0] == 0:
        return np.zeros((0, 4), np.int32), so
    key = _keys(cand, so)
    _, first = np.unique(key, return_index=True)
    return cand[first].astype(np.int32), so


def conv_features(feats, tab, w):
    """out[j] = sum_t W[t]^T . feats[tab[j, t]]   (float32; taps accumulated in order)."""
    ntaps = tab.shape[1]
    wt = w.reshape(ntaps, w.shape[-2], w.shape[-1]).astype(np.float32)
    out = np.zeros((tab.shape[0], w.shape[-1]), np.float32)
    for t in range(ntaps):
        idx = tab[:, t]
        m = idx >= 0
        if m.any():
            out[m] += feats[idx[m]].astype(np.float32) @ wt[t]
    return out


def subm_conv(feats, coords, shape, w):
    k = list(w.shape[:3])
    return conv_features(feats, neighbour_table(coords, shape, coords, k, [1, 1, 1], [0, 0, 0], True), w)


def sparse_conv(feats, coords, shape, w, stride, padding):
    k, s, p = list(w.shape[:3]), _triple(stride), _triple(padding)
    co, so = down_sites(coords, shape, k, s, p)
    return conv_features(feats, neighbour_table(coords, shape, co, k, s, p, False), w), co, so


def _bn_relu(x, sd, p, relu=True):
    y = (x - sd[p + ".running_mean"].numpy()) / np.sqrt(sd[p + ".running_var"].numpy() + BN_EPS) * sd[p + ".weight"].numpy() \
        + sd[p + ".bias"].numpy()
    y = y.astype(np.float32)
    return np.maximum(y, 0.0) if relu else y


def default_cfg():
    """The configuration streamingflow.py:111 builds (block_type 'basicblock')."""
    return dict(in_channels=5, sparse_shape=[1600, 1600, 41], output_channels=128, base_channels=16,
                encoder_channels=[[16, 16, 32], [32, 32, 64], [64, 64, 128], [128, 128]],
                encoder_paddings=[[0, 0, 1], [0, 0, 1], [0, 0, [1, 1, 0]], [0, 0]])


def sparse_encoder_forward(sd, voxel_features, coors, batch_size, cfg):
    """sparse_encoder.py:100-139, block_type 'basicblock', order (conv, norm, act).
    voxel_features [N, Cin] f32, coors [N, 4] int (batch, x, y, z) -> dense [B, C*D, H, W] float32 tensor."""
    feats = np.asarray(voxel_features, np.float32)
    coords = np.asarray(coors, np.int32)
    shape = list(cfg["sparse_shape"])
    npw = lambda k: sd[k].numpy()
    x = _bn_relu(subm_conv(feats, coords, shape, npw("conv_input.0.weight")), sd, "conv_input.1")
    n_stage = len(cfg["encoder_channels"])
    for i, blocks in enumerate(cfg["encoder_channels"]):
        for j in range(len(blocks)):
            p = f"encoder_layers.encoder_layer{i + 1}.{j}"
            if cfg.get("block_type", "basicblock") == "conv_module":        # sparse_encoder.py:166-179, :203-213
                if i != 0 and j == 0:
                    y, coords, shape = sparse_conv(x, coords, shape, npw(p + ".0.weight"), 2, cfg["encoder_paddings"][i][j])
                else:
                    y = subm_conv(x, coords, shape, npw(p + ".0.weight"))
                x = _bn_relu(y, sd, p + ".1")
            elif j == len(blocks) - 1 and i != n_stage - 1:      # strided SparseConv3d + BN + ReLU
                pad = cfg["encoder_paddings"][i][j]
                y, coords, shape = sparse_conv(x, coords, shape, npw(p + ".0.weight"), 2, pad)
                x = _bn_relu(y, sd, p + ".1")
            else:                                                 # SparseBasicBlock (sparse_block.py:88-107)
                y = _bn_relu(subm_conv(x, coords, shape, npw(p + ".conv1.weight")), sd, p + ".bn1")
                y = _bn_relu(subm_conv(y, coords, shape, npw(p + ".conv2.weight")), sd, p + ".bn2", relu=False)
                x = np.maximum(y + x, 0.0)
    y, coords, shape = sparse_conv(x, coords, shape, npw("conv_out.0.weight"), (1, 1, 2), 0)
    x = _bn_relu(y, sd, "conv_out.1")
    C = x.shape[1]
    dense = np.zeros((batch_size, C, shape[0], shape[1], shape[2]), np.float32)      # structure.py dense(): [B, C, *spatial]
    dense[coords[:, 0], :, coords[:, 1], coords[:, 2], coords[:, 3]] = x
    out = torch.from_numpy(dense).permute(0, 1, 4, 2, 3).contiguous()
    return out.view(batch_size, C * shape[2], shape[0], shape[1])


# ---- dense formulation (small grids): the same layers through torch.conv3d + activity masks ------------------
def _dense(feats, coords, shape, B):
    d = torch.zeros((B, feats.shape[1], *shape))
    m = torch.zeros((B, 1, *shape))
    c = torch.as_tensor(np.asarray(coords), dtype=torch.long)
    d[c[:, 0], :, c[:, 1], c[:, 2], c[:, 3]] = torch.as_tensor(np.asarray(feats), dtype=torch.float32)
    m[c[:, 0], 0, c[:, 1], c[:, 2], c[:, 3]] = 1.0
    return d, m


def dense_subm_conv(d, m, w):
    k = w.shape[:3]
    wt = torch.as_tensor(w).permute(4, 3, 0, 1, 2).contiguous()
    return F.conv3d(d, wt, padding=[k[0] // 2, k[1] // 2, k[2] // 2]) * m, m


def dense_sparse_conv(d, m, w, stride, padding):
    s, p = _triple(stride), _triple(padding)
    wt = torch.as_tensor(w).permute(4, 3, 0, 1, 2).contiguous()
    y = F.conv3d(d, wt, stride=s, padding=p)
    mo = (F.conv3d(m, torch.ones((1, 1, *w.shape[:3])), stride=s, padding=p) > 0).float()
    return y * mo, mo


def sparse_encoder_forward_dense(sd, voxel_features, coors, batch_size, cfg):
    def bn(y, m, p, relu=True):
        sh = (1, -1, 1, 1, 1)
        z = (y - sd[p + ".running_mean"].view(sh)) / torch.sqrt(sd[p + ".running_var"].view(sh) + BN_EPS) * sd[p + ".weight"].view(sh) \
            + sd[p + ".bias"].view(sh)
        return (F.relu(z) if relu else z) * m
    d, m = _dense(voxel_features, coors, cfg["sparse_shape"], batch_size)
    y, m = dense_subm_conv(d, m, sd["conv_input.0.weight"].numpy())
    x = bn(y, m, "conv_input.1")
    n_stage = len(cfg["encoder_channels"])
    for i, blocks in enumerate(cfg["encoder_channels"]):
        for j in range(len(blocks)):
            p = f"encoder_layers.encoder_layer{i + 1}.{j}"
            if cfg.get("block_type", "basicblock") == "conv_module":
                if i != 0 and j == 0:
                    y, m = dense_sparse_conv(x, m, sd[p + ".0.weight"].numpy(), 2, cfg["encoder_paddings"][i][j])
                else:
                    y, _ = dense_subm_conv(x, m, sd[p + ".0.weight"].numpy())
                x = bn(y, m, p + ".1")
            elif j == len(blocks) - 1 and i != n_stage - 1:
                y, m = dense_sparse_conv(x, m, sd[p + ".0.weight"].numpy(), 2, cfg["encoder_paddings"][i][j])
                x = bn(y, m, p + ".1")
            else:
                y, _ = dense_subm_conv(x, m, sd[p + ".conv1.weight"].numpy())
                y = bn(y, m, p + ".bn1")
                y, _ = dense_subm_conv(y, m, sd[p + ".conv2.weight"].numpy())
                y = bn(y, m, p + ".bn2", relu=False)
                x = F.relu(y + x) * m
    y, m = dense_sparse_conv(x, m, sd["conv_out.0.weight"].numpy(), (1, 1, 2), 0)
    x = bn(y, m, "conv_out.1")
    B, C, H, W, D = x.shape
    return x.permute(0, 1, 4, 2, 3).contiguous().view(B, C * D, H, W)


def state_dict_shapes(cfg):
    """Parameter / buffer names and shapes of the reference module (spconv weights are [kx,ky,kz,Cin,Cout];
    mmdet's BasicBlock registers its norms as bn1 / bn2)."""
    sh = {}

    def conv(name, k, cin, cout):
        sh[name + ".weight"] = (*k, cin, cout)

    def bn(name, c):
        for s in ("weight", "bias", "running_mean", "running_var"):
            sh[f"{name}.{s}"] = (c,)
        sh[name + ".num_batches_tracked"] = ()
    conv("conv_input.0", (3, 3, 3), cfg["in_channels"], cfg["base_channels"])
    bn("conv_input.1", cfg["base_channels"])
    cin = cfg["base_channels"]
    n_stage = len(cfg["encoder_channels"])
    for i, blocks in enumerate(cfg["encoder_channels"]):
        for j, cout in enumerate(blocks):
            p = f"encoder_layers.encoder_layer{i + 1}.{j}"
            if cfg.get("block_type", "basicblock") == "conv_module" or (j == len(blocks) - 1 and i != n_stage - 1):
                conv(p + ".0", (3, 3, 3), cin, cout)
                bn(p + ".1", cout)
            else:
                conv(p + ".conv1", (3, 3, 3), cout, cout)
                bn(p + ".bn1", cout)
                conv(p + ".conv2", (3, 3, 3), cout, cout)
                bn(p + ".bn2", cout)
            cin = cout
    conv("conv_out.0", (1, 1, 3), cin, cfg["output_channels"])
    bn("conv_out.1", cfg["output_channels"])
    return sh
