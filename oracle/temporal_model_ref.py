"""TEST INFRASTRUCTURE — CPU restatement of ``TemporalModel`` (second half).

Not imported by the product.  Follows streamingflow/models/temporal_model.py:8-69 and
streamingflow/layers/temporal.py:250-275 (CausalConv3d), :318-330 (1x1x1 conv + BN + ReLU), :394-432
(PyramidSpatioTemporalPooling), :435-490 (TemporalBlock); the final ``DeepLabHead`` is
``oracle.ref_torch.deeplab_head``.  Pinned: bit-identical to the reference class imported from
/root/reference (oracle/gen_golden.py --only temporal).  ``Bottleneck3D`` in-between layers
(n_spatial_layers_between_temporal_layers > 0; 0 in every shipped config) are restated too.
"""
import torch
import torch.nn.functional as F

from . import ref_torch as R


def _bn(x, sd, p):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"], False, 0.0, 1e-5)


def _c111(x, sd, p):
    """temporal.py:318-330."""
    return F.relu(_bn(F.conv3d(x, sd[p + ".conv.weight"]), sd, p + ".norm"))


def _causal(x, sd, p):
    """temporal.py:250-275: zero pad (kt-1) frames on the left, (k-1)//2 spatially; conv3d; BN; ReLU."""
    w = sd[p + ".conv.weight"]
    kt, kh, kw = w.shape[2:]
    x = F.pad(x, ((kw - 1) // 2, (kw - 1) // 2, (kh - 1) // 2, (kh - 1) // 2, kt - 1, 0))
    return F.relu(_bn(F.conv3d(x, w), sd, p + ".norm"))


def _pyramid(x, sd, p, pool_hw):
    """temporal.py:394-432 with pool_sizes = [(2, h, w)]."""
    b, _, t, h, w = x.shape
    ph, pw = pool_hw
    y = F.avg_pool3d(x, (2, ph, pw), stride=(1, ph, pw), padding=(1, 0, 0), count_include_pad=False)
    y = _c111(y, sd, p + ".features.0.conv_bn_relu")[:, :, :-1].contiguous()
    c = y.shape[1]
    y = F.interpolate(y.view(b * t, c, *y.shape[-2:]), (h, w), mode="bilinear", align_corners=False)
    return y.view(b, c, t, h, w)


def temporal_block(x, sd, p, pool_hw):
    """temporal.py:435-490.  x [b, C, T, H, W]."""
    paths = [_causal(_c111(x, sd, f"{p}.convolution_paths.{i}.0"), sd, f"{p}.convolution_paths.{i}.1") for i in (0, 1)]
    paths.append(_c111(x, sd, p + ".convolution_paths.2"))
    res = torch.cat(paths, dim=1)
    if (p + ".pyramid_pooling.features.0.conv_bn_relu.conv.weight") in sd:
        res = torch.cat([res, _pyramid(x, sd, p + ".pyramid_pooling", pool_hw)], dim=1)
    res = _c111(res, sd, p + ".aggregation.0")
    if (p + ".projection.0.weight") in sd:
        x = _bn(F.conv3d(x, sd[p + ".projection.0.weight"]), sd, p + ".projection.1")
    return x + res


def bottleneck3d(x, sd, p):
    """temporal.py:333-391 (in-between spatial layers)."""
    y = _c111(x, sd, p + ".layers.conv_down_project")
    y = _causal(y, sd, p + ".layers.conv")
    y = _c111(y, sd, p + ".layers.conv_up_project")
    if (p + ".projection.0.weight") in sd:
        x = _bn(F.conv3d(x, sd[p + ".projection.0.weight"]), sd, p + ".projection.1")
    return y + x


def temporal_model_forward(sd, x, input_shape):
    """temporal_model.py:51-69.  x [b, s, c, h, w] -> [b, s, c_out, h, w]."""
    x = x.permute(0, 2, 1, 3, 4)
    i = 0
    while any(k.startswith(f"model.{i}.") for k in sd):
        p = f"model.{i}"
        x = temporal_block(x, sd, p, input_shape) if (p + ".aggregation.0.conv.weight") in sd else bottleneck3d(x, sd, p)
        i += 1
    x = x.permute(0, 2, 1, 3, 4).contiguous()
    b, s, c, h, w = x.shape
    y = R.deeplab_head(sd, "final_conv", x.view(b * s, c, h, w))
    return y.view(b, s, -1, h, w)
