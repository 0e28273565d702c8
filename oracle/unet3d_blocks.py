"""CPU restatement of the two constructible blocks of segmentation/models/unet3d.py (test oracle): ConvD (:20-47) and
ConvU (:50-79) with 'bn' / 'gn' (GroupNorm(4, C)) / 'in' normalisation.  Quirk kept: ConvD computes a
conv2/bn2/relu(/dropout) branch and then overwrites it — the returned value is relu(x + bn3(conv3(x))) with
x = bn1(conv1(.)); conv2/bn2 parameters exist (state_dict) but never influence the output.
`Unet` itself raises in its constructor in the reference (unet3d.py:85) and is not restated."""
import torch
import torch.nn as nn
import torch.nn.functional as F


def normalization(planes, norm="gn"):
    if norm == "bn":
        return nn.BatchNorm3d(planes)
    if norm == "gn":
        return nn.GroupNorm(4, planes)
    if norm == "in":
        return nn.InstanceNorm3d(planes)
    raise ValueError("normalization type {} is not supported".format(norm))


class ConvD(nn.Module):
    def __init__(self, inplanes, planes, dropout=0.0, norm="gn", first=False):
        super().__init__()
        self.first, self.dropout = first, dropout
        self.maxpool = nn.MaxPool3d(2, 2)
        self.relu = nn.ReLU(inplace=True)
        self.conv1 = nn.Conv3d(inplanes, planes, 3, 1, 1, bias=False)
        self.bn1 = normalization(planes, norm)
        self.conv2 = nn.Conv3d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = normalization(planes, norm)
        self.conv3 = nn.Conv3d(planes, planes, 3, 1, 1, bias=False)
        self.bn3 = normalization(planes, norm)

    def forward(self, x):
        if not self.first:
            x = self.maxpool(x)
        x = self.bn1(self.conv1(x))
        return self.relu(x + self.bn3(self.conv3(x)))   # the conv2 branch of the reference is dead code


class ConvU(nn.Module):
    def __init__(self, planes, norm="gn", first=False):
        super().__init__()
        self.first = first
        if not first:
            self.conv1 = nn.Conv3d(2 * planes, planes, 3, 1, 1, bias=False)
            self.bn1 = normalization(planes, norm)
        self.conv2 = nn.Conv3d(planes, planes // 2, 1, 1, 0, bias=False)
        self.bn2 = normalization(planes // 2, norm)
        self.conv3 = nn.Conv3d(planes, planes, 3, 1, 1, bias=False)
        self.bn3 = normalization(planes, norm)
        self.relu = nn.ReLU(inplace=True)

    def forward(self, x, prev):
        if not self.first:
            x = self.relu(self.bn1(self.conv1(x)))
        y = F.interpolate(x, scale_factor=2, mode="trilinear", align_corners=False)
        y = self.relu(self.bn2(self.conv2(y)))
        y = torch.cat([prev, y], 1)
        return self.relu(self.bn3(self.conv3(y)))
