"""UNetRecon — pure-PyTorch CPU restatement of `unet.UNet` as the reference constructs it
(segmentation/routine.py:346-356: in_channels=1, out_classes=2, dimensions=3, num_encoding_blocks=3,
out_channels_first_layer=c0, normalization='batch', upsampling_type='linear', padding=True, activation='PReLU').

PARITY: topology pinned by strict-loading segmentation/weights/*.pth; forward semantics
(conv->BN->PReLU, MaxPool3d(2), trilinear x2 align_corners=False, cat((skip, up))) per Appendix A.2/A.3.
Functional parity with the upstream package itself: unpinned (source absent).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class _ConvBlock(nn.Module):
    def __init__(self, cin, cout, norm, act, k=3):
        super().__init__()
        layers = [nn.Conv3d(cin, cout, k, padding=k // 2)]
        self.conv_layer = layers[0]
        self.norm_layer = nn.BatchNorm3d(cout) if norm else None
        self.activation_layer = nn.PReLU() if act else None
        self.dropout_layer = None
        if self.norm_layer is not None:
            layers.append(self.norm_layer)
        if self.activation_layer is not None:
            layers.append(self.activation_layer)
        self.block = nn.Sequential(*layers)

    def forward(self, x):
        return self.block(x)


class _EncBlock(nn.Module):
    def __init__(self, cin, f, first, pool):
        super().__init__()
        self.conv1 = _ConvBlock(cin, f, norm=not first, act=True)
        self.conv2 = _ConvBlock(f, 2 * f, norm=True, act=True)
        self.downsample = nn.MaxPool3d(2) if pool else None

    def forward(self, x):
        x = self.conv2(self.conv1(x))
        return (self.downsample(x), x) if self.downsample is not None else x


class _Encoder(nn.Module):
    def __init__(self, cin, f, nblocks):
        super().__init__()
        self.encoding_blocks = nn.ModuleList()
        for i in range(nblocks):
            self.encoding_blocks.append(_EncBlock(cin, f, first=i == 0, pool=True))
            cin = 2 * f
            f = cin

    def forward(self, x):
        skips = []
        for b in self.encoding_blocks:
            x, s = b(x)
            skips.append(s)
        return skips, x


class _DecBlock(nn.Module):
    def __init__(self, skip_c, skip_first=True):
        super().__init__()
        self.skip_first = skip_first
        self.conv1 = _ConvBlock(3 * skip_c, skip_c, norm=True, act=True)
        self.conv2 = _ConvBlock(skip_c, skip_c, norm=True, act=True)

    def forward(self, skip, x):
        x = F.interpolate(x, scale_factor=2, mode="trilinear", align_corners=False)
        x = torch.cat((skip, x) if self.skip_first else (x, skip), dim=1)
        return self.conv2(self.conv1(x))


class _Decoder(nn.Module):
    def __init__(self, skip_c, nblocks, skip_first):
        super().__init__()
        self.decoding_blocks = nn.ModuleList()
        for _ in range(nblocks):
            self.decoding_blocks.append(_DecBlock(skip_c, skip_first))
            skip_c //= 2

    def forward(self, skips, x):
        for s, b in zip(reversed(skips), self.decoding_blocks):
            x = b(s, x)
        return x


class UNetRecon(nn.Module):
    def __init__(self, in_channels=1, out_classes=2, num_encoding_blocks=3, out_channels_first_layer=8, skip_first=True):
        super().__init__()
        depth = num_encoding_blocks - 1
        c0 = out_channels_first_layer
        self.encoder = _Encoder(in_channels, c0, depth)
        c = c0 * 2 ** depth
        self.bottom_block = _EncBlock(c, c, first=False, pool=False)
        self.decoder = _Decoder(c, depth, skip_first)
        self.classifier = _ConvBlock(2 * c0, out_classes, norm=False, act=False, k=1)

    def forward(self, x):
        skips, x = self.encoder(x)
        x = self.bottom_block(x)
        x = self.decoder(skips, x)
        return self.classifier(x)
