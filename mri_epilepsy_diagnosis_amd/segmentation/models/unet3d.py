"""MI355X-native drop-in for the constructible parts of segmentation/models/unet3d.py: `normalization`, `ConvD`
(:20-47) and `ConvU` (:50-79) — same constructor arguments, attribute names (state_dict keys) and outputs, including
the reference's dead conv2 branch in ConvD (its parameters are kept, its compute is not run: it never reaches the
output).  BatchNorm3d / GroupNorm(4, C) / InstanceNorm3d + ReLU run as fused HIP passes; `Unet` is not provided because
the reference's class raises in its own constructor (unet3d.py:85)."""
import torch.nn as tnn

from ... import nn as mnn
from ... import ops


def normalization(planes, norm="gn"):
    if norm == "bn":
        return mnn.BatchNorm3d(planes)
    if norm == "gn":
        return mnn.GroupNorm(4, planes)
    if norm == "in":
        return mnn.InstanceNorm3d(planes)
    raise ValueError("normalization type {} is not supported".format(norm))


def _conv(cin, cout, k):
    return mnn.Conv3d(cin, cout, k, 1, k // 2, bias=False)


class ConvD(tnn.Module):
    def __init__(self, inplanes, planes, dropout=0.0, norm="gn", first=False):
        super().__init__()
        self.first = first
        self.maxpool = mnn.MaxPool3d(2, 2)
        self.dropout = dropout
        self.relu = mnn.ReLU(inplace=True)
        self.conv1, self.bn1 = _conv(inplanes, planes, 3), normalization(planes, norm)
        self.conv2, self.bn2 = _conv(planes, planes, 3), normalization(planes, norm)
        self.conv3, self.bn3 = _conv(planes, planes, 3), normalization(planes, norm)

    def forward(self, x):
        if not self.first:
            x = self.maxpool(x)
        x = mnn.fused_norm_act(self.bn1, None, self.conv1(x))
        y = mnn.fused_norm_act(self.bn3, None, self.conv3(x))
        return self.relu(ops.add(x, y))


class ConvU(tnn.Module):
    def __init__(self, planes, norm="gn", first=False):
        super().__init__()
        self.first = first
        if not self.first:
            self.conv1, self.bn1 = _conv(2 * planes, planes, 3), normalization(planes, norm)
        self.conv2, self.bn2 = _conv(planes, planes // 2, 1), normalization(planes // 2, norm)
        self.conv3, self.bn3 = _conv(planes, planes, 3), normalization(planes, norm)
        self.relu = mnn.ReLU(inplace=True)

    def forward(self, x, prev):
        if not self.first:
            x = mnn.fused_norm_act(self.bn1, self.relu, self.conv1(x))
        y = ops.upsample3d(x, scale_factor=2, mode="trilinear", align_corners=False)
        y = mnn.fused_norm_act(self.bn2, self.relu, self.conv2(y))
        y = ops.cat_channels([prev, y])
        return mnn.fused_norm_act(self.bn3, self.relu, self.conv3(y))
