"""MI355X-native drop-in for segmentation/models/modified_3dunet.py::Modified3DUNet(in_channels, n_classes,
base_n_filter): same attribute names / state_dict keys, same forward (modified_3dunet.py:97-189) including the
double application of each ``norm_lrelu_conv_cK`` (shared weights) and Dropout3d(0.6) in train mode.

InstanceNorm3d + LeakyReLU always run as one fused HIP pass (per-(n,c) statistics by wave shuffles + LDS), residual
sums and concats are channel-slice kernels, nearest x2 upsampling is a gather.
"""
import torch.nn as tnn

from ... import nn as mnn
from ... import ops


def _c3(cin, cout, stride=1):
    return mnn.Conv3d(cin, cout, kernel_size=3, stride=stride, padding=1, bias=False)


def _c1(cin, cout):
    return mnn.Conv3d(cin, cout, kernel_size=1, stride=1, padding=0, bias=False)


class Modified3DUNet(tnn.Module):
    def __init__(self, in_channels, n_classes, base_n_filter=8):
        super().__init__()
        self.in_channels = in_channels
        self.n_classes = n_classes
        self.base_n_filter = base_n_filter
        b = base_n_filter

        self.lrelu = mnn.LeakyReLU()
        self.dropout3d = mnn.Dropout3d(p=0.6)
        self.upsacle = mnn.Upsample(scale_factor=2, mode="nearest")
        self.softmax = tnn.Softmax(dim=1)

        self.conv3d_c1_1 = _c3(in_channels, b)
        self.conv3d_c1_2 = _c3(b, b)
        self.lrelu_conv_c1 = self.lrelu_conv(b, b)
        self.inorm3d_c1 = mnn.InstanceNorm3d(b)
        for level in (2, 3, 4):
            width = b * 2 ** (level - 1)
            setattr(self, "conv3d_c%d" % level, _c3(width // 2, width, stride=2))
            setattr(self, "norm_lrelu_conv_c%d" % level, self.norm_lrelu_conv(width, width))
            setattr(self, "inorm3d_c%d" % level, mnn.InstanceNorm3d(width))
        self.conv3d_c5 = _c3(b * 8, b * 16, stride=2)
        self.norm_lrelu_conv_c5 = self.norm_lrelu_conv(b * 16, b * 16)
        self.norm_lrelu_upscale_conv_norm_lrelu_l0 = self.norm_lrelu_upscale_conv_norm_lrelu(b * 16, b * 8)
        self.conv3d_l0 = _c1(b * 8, b * 8)
        self.inorm3d_l0 = mnn.InstanceNorm3d(b * 8)

        for level, width in ((1, b * 16), (2, b * 8), (3, b * 4)):
            setattr(self, "conv_norm_lrelu_l%d" % level, self.conv_norm_lrelu(width, width))
            setattr(self, "conv3d_l%d" % level, _c1(width, width // 2))
            setattr(self, "norm_lrelu_upscale_conv_norm_lrelu_l%d" % level,
                    self.norm_lrelu_upscale_conv_norm_lrelu(width // 2, width // 4))
        self.conv_norm_lrelu_l4 = self.conv_norm_lrelu(b * 2, b * 2)
        self.conv3d_l4 = _c1(b * 2, n_classes)
        self.ds2_1x1_conv3d = _c1(b * 8, n_classes)
        self.ds3_1x1_conv3d = _c1(b * 4, n_classes)

    def conv_norm_lrelu(self, feat_in, feat_out):
        return mnn.FusedSequential(_c3(feat_in, feat_out), mnn.InstanceNorm3d(feat_out), mnn.LeakyReLU())

    def norm_lrelu_conv(self, feat_in, feat_out):
        return mnn.FusedSequential(mnn.InstanceNorm3d(feat_in), mnn.LeakyReLU(), _c3(feat_in, feat_out))

    def lrelu_conv(self, feat_in, feat_out):
        return mnn.FusedSequential(mnn.LeakyReLU(), _c3(feat_in, feat_out))

    def norm_lrelu_upscale_conv_norm_lrelu(self, feat_in, feat_out):
        return mnn.FusedSequential(mnn.InstanceNorm3d(feat_in), mnn.LeakyReLU(),
                                   mnn.Upsample(scale_factor=2, mode="nearest"), _c3(feat_in, feat_out),
                                   mnn.InstanceNorm3d(feat_out), mnn.LeakyReLU())

    def forward(self, x):
        out = self.conv3d_c1_1(x)
        residual = out
        out = self.conv3d_c1_2(self.lrelu(out))
        out = self.lrelu_conv_c1(self.dropout3d(out))
        out = ops.add(out, residual)
        contexts = [self.lrelu(out)]
        out = mnn.fused_norm_act(self.inorm3d_c1, self.lrelu, out)

        for level in (2, 3, 4):
            out = getattr(self, "conv3d_c%d" % level)(out)
            residual = out
            shared = getattr(self, "norm_lrelu_conv_c%d" % level)
            out = shared(self.dropout3d(shared(out)))
            out = ops.add(out, residual)
            out = mnn.fused_norm_act(getattr(self, "inorm3d_c%d" % level), self.lrelu, out)
            contexts.append(out)

        out = self.conv3d_c5(out)
        residual = out
        out = self.norm_lrelu_conv_c5(self.dropout3d(self.norm_lrelu_conv_c5(out)))
        out = ops.add(out, residual)
        out = self.norm_lrelu_upscale_conv_norm_lrelu_l0(out)
        out = mnn.fused_norm_act(self.inorm3d_l0, self.lrelu, self.conv3d_l0(out))

        deep = {}
        for level in (1, 2, 3):
            out = ops.cat_channels([out, contexts[4 - level]])
            out = getattr(self, "conv_norm_lrelu_l%d" % level)(out)
            deep[level] = out
            out = getattr(self, "conv3d_l%d" % level)(out)
            out = getattr(self, "norm_lrelu_upscale_conv_norm_lrelu_l%d" % level)(out)

        out = ops.cat_channels([out, contexts[0]])
        out = self.conv_norm_lrelu_l4(out)
        out_pred = self.conv3d_l4(out)

        ds2_up = self.upsacle(self.ds2_1x1_conv3d(deep[2]))
        ds23_up = self.upsacle(ops.add(ds2_up, self.ds3_1x1_conv3d(deep[3])))
        return ops.add(out_pred, ds23_up)
