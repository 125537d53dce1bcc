"""TEST INFRASTRUCTURE — CPU restatement of the BEV ``Decoder`` (first half).

Not imported by the product.  Follows streamingflow/models/decoder.py:8-140 and
streamingflow/layers/convolutions.py:204-215 (``UpsamplingAdd``).  The backbone blocks come from a
third-party dependency that is absent here and in /root/reference: ``torchvision.models.resnet.resnet18``
(version not pinned by the reference — no requirements file names it; ``README.md:54`` defers to the
ST-P3 / BEVFusion environments).  Its published ``BasicBlock`` (conv3x3(stride) - BN - ReLU - conv3x3 - BN,
identity or conv1x1(stride)+BN shortcut, ReLU after the sum; layer1: 2 x 64, layer2: 2 x 128 stride 2,
layer3: 2 x 256 stride 2) is restated in ``tv_resnet18`` with torchvision's parameter names.

Parity status: the fixtures (tests/golden/decoder.npz) run the REFERENCE's Decoder class (imported
from /root/reference) on top of ``tv_resnet18`` — decoder.py itself is pinned, the torchvision blocks
are pinned only to this restatement ("parity partially unpinned").
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class BasicBlock(nn.Module):
    """torchvision.models.resnet.BasicBlock (expansion 1), parameter names as published."""

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=1, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        identity = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        if self.downsample is not None:
            identity = self.downsample(x)
        return self.relu(out + identity)


def _make_layer(inplanes, planes, blocks, stride):
    down = None
    if stride != 1 or inplanes != planes:
        down = nn.Sequential(nn.Conv2d(inplanes, planes, 1, stride=stride, bias=False), nn.BatchNorm2d(planes))
    layers = [BasicBlock(inplanes, planes, stride, down)] + [BasicBlock(planes, planes) for _ in range(1, blocks)]
    return nn.Sequential(*layers)


class _ResNet18Trunk(nn.Module):
    """The attributes decoder.py:22-31 takes from ``resnet18(pretrained=False, zero_init_residual=True)``."""

    def __init__(self, zero_init_residual=True):
        super().__init__()
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.layer1 = _make_layer(64, 64, 2, 1)
        self.layer2 = _make_layer(64, 128, 2, 2)
        self.layer3 = _make_layer(128, 256, 2, 2)
        if zero_init_residual:
            for m in self.modules():
                if isinstance(m, BasicBlock):
                    nn.init.constant_(m.bn2.weight, 0)


def tv_resnet18(pretrained=False, zero_init_residual=False, **kw):
    return _ResNet18Trunk(zero_init_residual)


# ---- functional restatement over the reference's state_dict keys -----------------------------------------
def _bn(x, sd, p):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"], False, 0.0, 1e-5)


def _block(x, sd, p, stride):
    out = F.relu(_bn(F.conv2d(x, sd[p + ".conv1.weight"], None, stride, 1), sd, p + ".bn1"))
    out = _bn(F.conv2d(out, sd[p + ".conv2.weight"], None, 1, 1), sd, p + ".bn2")
    idt = x
    if (p + ".downsample.0.weight") in sd:
        idt = _bn(F.conv2d(x, sd[p + ".downsample.0.weight"], None, stride, 0), sd, p + ".downsample.1")
    return F.relu(out + idt)


def _up_add(x, skip, sd, p):
    """convolutions.py:204-215."""
    x = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False)
    x = _bn(F.conv2d(x, sd[p + ".upsample_layer.1.weight"]), sd, p + ".upsample_layer.2")
    return x + skip


def _head(x, sd, p, sigmoid=False):
    y = F.relu(_bn(F.conv2d(x, sd[p + ".0.weight"], None, 1, 1), sd, p + ".1"))
    y = F.conv2d(y, sd[p + ".3.weight"], sd[p + ".3.bias"])
    return torch.sigmoid(y) if sigmoid else y


HEADS = (("segmentation", "segmentation_head", False), ("pedestrian", "pedestrian_head", False), ("hdmap", "hdmap_head", False),
         ("instance_center", "instance_center_head", True), ("instance_offset", "instance_offset_head", False),
         ("instance_flow", "instance_future_head", False), ("costvolume", "costvolume_head", False))


def decoder_forward(sd, x, n_present):
    """decoder.py:93-140.  x [b, s, c, h, w] -> dict of head outputs (None for heads absent from ``sd``)."""
    b, s, c, h, w = x.shape
    x = x.view(b * s, c, h, w)
    skip1 = x
    x = F.relu(_bn(F.conv2d(x, sd["first_conv.weight"], None, 2, 3), sd, "bn1"))
    x = _block(_block(x, sd, "layer1.0", 1), sd, "layer1.1", 1)
    skip2 = x
    x = _block(_block(x, sd, "layer2.0", 2), sd, "layer2.1", 1)
    skip3 = x
    x = _block(_block(x, sd, "layer3.0", 2), sd, "layer3.1", 1)
    x = _up_add(x, skip3, sd, "up3_skip")
    x = _up_add(x, skip2, sd, "up2_skip")
    x = _up_add(x, skip1, sd, "up1_skip")
    out = {}
    for name, p, sig in HEADS:
        if (p + ".0.weight") not in sd:
            out[name] = None
            continue
        if name == "hdmap":
            out[name] = _head(x.view(b, s, *x.shape[1:])[:, n_present - 1], sd, p)
            continue
        y = _head(x, sd, p, sig)
        if name == "costvolume":
            y = y.squeeze(1)
        out[name] = y.view(b, s, *y.shape[1:])
    return out
