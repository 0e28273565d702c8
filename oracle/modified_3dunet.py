"""CPU restatement of segmentation/models/modified_3dunet.py::Modified3DUNet (test oracle): Isensee-style 3-D
U-Net — stride-2 context convs, InstanceNorm3d + LeakyReLU, nearest x2 upsampling, deep supervision.
Attribute names equal the reference's so state_dicts interchange.  Quirks kept: each
``norm_lrelu_conv_cK`` module is applied TWICE per level (shared weights, modified_3dunet.py:114-116), and level 1
uses ``lrelu_conv_c1`` after Dropout3d(0.6).
"""
import torch
import torch.nn as nn


def _conv3(cin, cout, stride=1):
    return nn.Conv3d(cin, cout, kernel_size=3, stride=stride, padding=1, bias=False)


def _conv1(cin, cout):
    return nn.Conv3d(cin, cout, kernel_size=1, stride=1, padding=0, bias=False)


class Modified3DUNet(nn.Module):
    def __init__(self, in_channels, n_classes, base_n_filter=8):
        super().__init__()
        self.in_channels, self.n_classes, self.base_n_filter = in_channels, n_classes, base_n_filter
        b = base_n_filter
        self.lrelu = nn.LeakyReLU()
        self.dropout3d = nn.Dropout3d(p=0.6)
        self.upsacle = nn.Upsample(scale_factor=2, mode="nearest")
        self.softmax = nn.Softmax(dim=1)
        # context pathway (construction order = reference order, so seeded init matches)
        self.conv3d_c1_1 = _conv3(in_channels, b)
        self.conv3d_c1_2 = _conv3(b, b)
        self.lrelu_conv_c1 = nn.Sequential(nn.LeakyReLU(), _conv3(b, b))
        self.inorm3d_c1 = nn.InstanceNorm3d(b)
        for lvl, mult in ((2, 2), (3, 4), (4, 8)):
            setattr(self, "conv3d_c%d" % lvl, _conv3(b * mult // 2, b * mult, stride=2))
            setattr(self, "norm_lrelu_conv_c%d" % lvl, self._norm_lrelu_conv(b * mult, b * mult))
            setattr(self, "inorm3d_c%d" % lvl, nn.InstanceNorm3d(b * mult))
        self.conv3d_c5 = _conv3(b * 8, b * 16, stride=2)
        self.norm_lrelu_conv_c5 = self._norm_lrelu_conv(b * 16, b * 16)
        self.norm_lrelu_upscale_conv_norm_lrelu_l0 = self._up_block(b * 16, b * 8)
        self.conv3d_l0 = _conv1(b * 8, b * 8)
        self.inorm3d_l0 = nn.InstanceNorm3d(b * 8)
        # localisation pathway
        self.conv_norm_lrelu_l1 = self._conv_norm_lrelu(b * 16, b * 16)
        self.conv3d_l1 = _conv1(b * 16, b * 8)
        self.norm_lrelu_upscale_conv_norm_lrelu_l1 = self._up_block(b * 8, b * 4)
        self.conv_norm_lrelu_l2 = self._conv_norm_lrelu(b * 8, b * 8)
        self.conv3d_l2 = _conv1(b * 8, b * 4)
        self.norm_lrelu_upscale_conv_norm_lrelu_l2 = self._up_block(b * 4, b * 2)
        self.conv_norm_lrelu_l3 = self._conv_norm_lrelu(b * 4, b * 4)
        self.conv3d_l3 = _conv1(b * 4, b * 2)
        self.norm_lrelu_upscale_conv_norm_lrelu_l3 = self._up_block(b * 2, b)
        self.conv_norm_lrelu_l4 = self._conv_norm_lrelu(b * 2, b * 2)
        self.conv3d_l4 = _conv1(b * 2, n_classes)
        self.ds2_1x1_conv3d = _conv1(b * 8, n_classes)
        self.ds3_1x1_conv3d = _conv1(b * 4, n_classes)

    @staticmethod
    def _conv_norm_lrelu(cin, cout):
        return nn.Sequential(_conv3(cin, cout), nn.InstanceNorm3d(cout), nn.LeakyReLU())

    @staticmethod
    def _norm_lrelu_conv(cin, cout):
        return nn.Sequential(nn.InstanceNorm3d(cin), nn.LeakyReLU(), _conv3(cin, cout))

    @staticmethod
    def _up_block(cin, cout):
        return nn.Sequential(nn.InstanceNorm3d(cin), nn.LeakyReLU(), nn.Upsample(scale_factor=2, mode="nearest"),
                             _conv3(cin, cout), nn.InstanceNorm3d(cout), nn.LeakyReLU())

    def forward(self, x):
        out = self.conv3d_c1_1(x)
        res = out
        out = self.conv3d_c1_2(self.lrelu(out))
        out = self.lrelu_conv_c1(self.dropout3d(out))
        out = out + res
        context = [self.lrelu(out)]
        out = self.lrelu(self.inorm3d_c1(out))
        for lvl in (2, 3, 4):
            out = getattr(self, "conv3d_c%d" % lvl)(out)
            res = out
            nlc = getattr(self, "norm_lrelu_conv_c%d" % lvl)
            out = nlc(self.dropout3d(nlc(out)))
            out = out + res
            out = self.lrelu(getattr(self, "inorm3d_c%d" % lvl)(out))
            context.append(out)
        out = self.conv3d_c5(out)
        res = out
        out = self.norm_lrelu_conv_c5(self.dropout3d(self.norm_lrelu_conv_c5(out)))
        out = out + res
        out = self.norm_lrelu_upscale_conv_norm_lrelu_l0(out)
        out = self.lrelu(self.inorm3d_l0(self.conv3d_l0(out)))
        ds = {}
        for lvl in (1, 2, 3):
            out = torch.cat([out, context[4 - lvl]], dim=1)
            out = getattr(self, "conv_norm_lrelu_l%d" % lvl)(out)
            ds[lvl] = out
            out = getattr(self, "conv3d_l%d" % lvl)(out)
            out = getattr(self, "norm_lrelu_upscale_conv_norm_lrelu_l%d" % lvl)(out)
        out = torch.cat([out, context[0]], dim=1)
        out = self.conv_norm_lrelu_l4(out)
        out_pred = self.conv3d_l4(out)
        d2 = self.upsacle(self.ds2_1x1_conv3d(ds[2]))
        d3 = self.upsacle(d2 + self.ds3_1x1_conv3d(ds[3]))
        return out_pred + d3
