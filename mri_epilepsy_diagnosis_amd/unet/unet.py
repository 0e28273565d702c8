"""3-D ``UNet`` with the constructor, forward signature and ``state_dict`` schema of the PyPI package ``unet``
(F. Pérez-García) that the reference trains: ``UNet(in_channels=1, out_classes=2, dimensions=3,
num_encoding_blocks=3, out_channels_first_layer=c0, normalization='batch', upsampling_type='linear', padding=True,
activation='PReLU')`` — segmentation/routine.py:346-356.

The package source is not part of the reference tree; the topology is pinned by the 18 shipped checkpoints
(SURVEY.md Appendix A): every ``ConvolutionalBlock`` registers its layers twice (``conv_layer``/``norm_layer``/
``activation_layer`` attributes and a ``block`` Sequential holding the same modules), the first conv of the first
encoder block has no normalisation, 3-D encoder blocks go in->f->2f, decoder blocks take cat(skip, upsampled).

Execution is MI355X-native: conv -> (BatchNorm+PReLU fused) per block, all through the HIP operators in ``ops``.
"""
import torch
import torch.nn as tnn

from .. import nn as mnn
from .. import ops

_UPSAMPLING = {"nearest": "nearest", "linear": "trilinear", "trilinear": "trilinear"}


class ConvolutionalBlock(tnn.Module):
    def __init__(self, dimensions, in_channels, out_channels, normalization=None, kernel_size=3, activation="ReLU",
                 preactivation=False, padding=0, padding_mode="zeros", dilation=None, dropout=0):
        super().__init__()
        if dimensions != 3:
            raise NotImplementedError("only the 3-D U-Net of the reference is implemented")
        if preactivation or padding_mode != "zeros" or dropout:
            raise NotImplementedError("preactivation / non-zero padding_mode / dropout are not used by the reference")
        dilation = 1 if dilation is None else dilation
        if padding:
            total = kernel_size + 2 * (dilation - 1) - 1
            padding = total // 2
        conv = mnn.Conv3d(in_channels, out_channels, kernel_size, padding=padding, dilation=dilation)
        norm = None
        if normalization is not None:
            norm = {"batch": mnn.BatchNorm3d, "instance": mnn.InstanceNorm3d}[normalization.lower()](out_channels)
        act = None
        if activation is not None:
            act = {"PReLU": mnn.PReLU, "ReLU": mnn.ReLU, "LeakyReLU": mnn.LeakyReLU}[activation]()
        self.conv_layer = conv
        self.norm_layer = norm
        self.activation_layer = act
        self.dropout_layer = None
        self.block = tnn.Sequential(*[m for m in (conv, norm, act) if m is not None])

    def forward(self, x, out=None):
        return mnn.conv_norm_act(self.conv_layer, self.norm_layer, self.activation_layer, x, out)


class EncodingBlock(tnn.Module):
    def __init__(self, in_channels, out_channels_first, dimensions, normalization, pooling_type, preactivation=False,
                 is_first_block=False, residual=False, padding=0, padding_mode="zeros", activation="ReLU",
                 dilation=None, dropout=0):
        super().__init__()
        if residual:
            raise NotImplementedError("residual encoding blocks are not used by the reference")
        self.conv1 = ConvolutionalBlock(dimensions, in_channels, out_channels_first,
                                        normalization=None if is_first_block else normalization,
                                        padding=padding, activation=activation, dilation=dilation)
        out_channels_second = 2 * out_channels_first  # 3-D rule
        self.conv2 = ConvolutionalBlock(dimensions, out_channels_first, out_channels_second, normalization=normalization,
                                        padding=padding, activation=activation, dilation=dilation)
        self.downsample = None
        if pooling_type is not None:
            if pooling_type != "max":
                raise NotImplementedError("only max pooling is used by the reference")
            self.downsample = mnn.MaxPool3d(kernel_size=2)

    def forward(self, x, skip_out=None):
        # skip_out = (concat buffer, 0): the block's output is written straight into the decoder's concat buffer
        x = self.conv2(self.conv1(x), skip_out)
        if self.downsample is None:
            return x
        # (pooled, skip) from one autograd node: the two gradients of x are summed inside the pool-backward kernel
        return self.downsample.forward_with_skip(x)

    @property
    def out_channels(self):
        return self.conv2.conv_layer.out_channels


class Encoder(tnn.Module):
    def __init__(self, in_channels, out_channels_first, dimensions, pooling_type, num_encoding_blocks, normalization,
                 padding=0, activation="ReLU", initial_dilation=None):
        super().__init__()
        self.encoding_blocks = tnn.ModuleList()
        self.dilation = initial_dilation
        first = True
        for _ in range(num_encoding_blocks):
            blk = EncodingBlock(in_channels, out_channels_first, dimensions, normalization, pooling_type,
                                is_first_block=first, padding=padding, activation=activation, dilation=self.dilation)
            first = False
            self.encoding_blocks.append(blk)
            in_channels = 2 * out_channels_first
            out_channels_first = in_channels
            if self.dilation is not None:
                self.dilation *= 2

    def forward(self, x, cat_bufs=None):
        skips = []
        for i, blk in enumerate(self.encoding_blocks):
            x, skip = blk(x, None if cat_bufs is None else (cat_bufs[i], 0))
            skips.append(skip)
        return skips, x

    @property
    def out_channels(self):
        return self.encoding_blocks[-1].out_channels


class DecodingBlock(tnn.Module):
    def __init__(self, in_channels_skip_connection, dimensions, upsampling_type, normalization, padding=0,
                 activation="ReLU", dilation=None):
        super().__init__()
        if upsampling_type == "conv":
            c = 2 * in_channels_skip_connection
            self.upsample = mnn.ConvTranspose3d(c, c, kernel_size=2, stride=2)
        else:
            mode = _UPSAMPLING[upsampling_type]
            self.upsample = mnn.Upsample(scale_factor=2, mode=mode, align_corners=False if mode == "trilinear" else None)
        in_first = in_channels_skip_connection * 3  # skip + 2*skip (3-D)
        self.conv1 = ConvolutionalBlock(dimensions, in_first, in_channels_skip_connection, normalization=normalization,
                                        padding=padding, activation=activation, dilation=dilation)
        self.conv2 = ConvolutionalBlock(dimensions, in_channels_skip_connection, in_channels_skip_connection,
                                        normalization=normalization, padding=padding, activation=activation,
                                        dilation=dilation)

    def forward(self, skip, x, cat_buf=None):
        if cat_buf is not None and isinstance(self.upsample, mnn.Upsample):
            # copy-free torch.cat((skip, up)) through a shared buffer: `skip` already lives in cat_buf[:, :Cs]; the upsample kernel
            # writes cat_buf[:, Cs:]; join_channels only ties the two producers together for autograd (skip first: SURVEY A.3)
            up = ops.upsample3d(x, self.upsample.size, self.upsample.scale_factor, self.upsample.mode,
                                self.upsample.align_corners, out=(cat_buf, skip.shape[1]))
            return self.conv2(self.conv1(ops.join_channels(cat_buf, [skip, up])))
        x = self.upsample(x)
        if skip.shape[2:] != x.shape[2:]:
            raise NotImplementedError("padding=False (centre-cropped skips) is not used by the reference")
        # cat((skip, x)) is never formed: the first convolution reads the two dense tensors (ops.conv3d_cat) and hands back two
        # dense gradients.  Slice writes into a shared 3C-channel buffer cost 1.5x (fp32) - 3.6x (bf16) of dense writes.
        return self.conv2(self.conv1((skip, x)))


class Decoder(tnn.Module):
    def __init__(self, in_channels_skip_connection, dimensions, upsampling_type, num_decoding_blocks, normalization,
                 padding=0, activation="ReLU", initial_dilation=None):
        super().__init__()
        self.decoding_blocks = tnn.ModuleList()
        self.dilation = initial_dilation
        for _ in range(num_decoding_blocks):
            self.decoding_blocks.append(DecodingBlock(in_channels_skip_connection, dimensions, upsampling_type,
                                                      normalization, padding=padding, activation=activation,
                                                      dilation=self.dilation))
            in_channels_skip_connection //= 2
            if self.dilation is not None:
                self.dilation //= 2

    def forward(self, skips, x, cat_bufs=None):
        for j, (skip, blk) in enumerate(zip(reversed(skips), self.decoding_blocks)):
            x = blk(skip, x, None if cat_bufs is None else cat_bufs[len(skips) - 1 - j])
        return x


class UNet(tnn.Module):
    def __init__(self, in_channels=1, out_classes=2, dimensions=2, num_encoding_blocks=5, out_channels_first_layer=64,
                 normalization=None, pooling_type="max", upsampling_type="conv", preactivation=False, residual=False,
                 padding=0, padding_mode="zeros", activation="ReLU", initial_dilation=None, dropout=0,
                 monte_carlo_dropout=0):
        super().__init__()
        if dimensions != 3:
            raise NotImplementedError("only dimensions=3 (the reference's configuration) is implemented")
        if preactivation or residual or dropout or monte_carlo_dropout or padding_mode != "zeros":
            raise NotImplementedError("option not used by the reference and not implemented")
        if not padding:
            raise NotImplementedError("padding=False (valid convolutions + cropped skips) is not used by the reference")
        depth = num_encoding_blocks - 1
        self.encoder = Encoder(in_channels, out_channels_first_layer, dimensions, pooling_type, depth, normalization,
                               padding=padding, activation=activation, initial_dilation=initial_dilation)
        c = self.encoder.out_channels if depth > 0 else in_channels
        first = c if depth > 0 else out_channels_first_layer
        self.bottom_block = EncodingBlock(c, first, dimensions, normalization, pooling_type=None, padding=padding,
                                          activation=activation, dilation=self.encoder.dilation,
                                          is_first_block=depth == 0)
        skip_channels = out_channels_first_layer * 2 ** depth
        self.decoder = Decoder(skip_channels, dimensions, upsampling_type, depth, normalization, padding=padding,
                               activation=activation, initial_dilation=self.encoder.dilation)
        self.monte_carlo_layer = None
        self.shared_concat_buffers = False   # round 1's concat-buffer scheme (A/B switch; not part of the reference's API)
        self.classifier = ConvolutionalBlock(dimensions, 2 * out_channels_first_layer, out_classes, kernel_size=1,
                                             activation=None)

    def forward(self, x):
        # Round 1's scheme (kept for A/B: set `shared_concat_buffers`): one NDHWC buffer per level holds cat((skip, upsampled)) and
        # both producers write their channel slice in place.  Default since round 2: no concatenation at all (DecodingBlock.forward).
        cat_bufs = None
        if self.shared_concat_buffers and x.is_cuda and len(self.encoder.encoding_blocks) and all(
                isinstance(b.upsample, mnn.Upsample) for b in self.decoder.decoding_blocks):
            cat_bufs = []
            n, sp = x.shape[0], tuple(x.shape[2:])
            for blk in self.encoder.encoding_blocks:
                if any(s % 2 for s in sp):
                    cat_bufs = None
                    break
                cat_bufs.append(ops.new_cat_buffer(n, 3 * blk.out_channels, sp, x))
                sp = tuple(s // 2 for s in sp)
        skips, enc = self.encoder(x, cat_bufs)
        enc = self.bottom_block(enc)
        x = self.decoder(skips, enc, cat_bufs)
        return self.classifier(x)
