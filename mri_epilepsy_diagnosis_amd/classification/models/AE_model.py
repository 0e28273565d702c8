"""MI355X-native drop-in for the reference's classification/models/AE_model.py: DownBlock, UpBlock, Encoder,
Decoder, AE, Discriminator, Classificator — same constructor kwargs, same ``forward`` signatures / return types
(``DownBlock.forward -> (x, shape_before_pool)``, ``Encoder.forward -> (x, size_list)``), same ModuleDict keys (so
encoder_93_6_4.pth / clf_93_6_4.pth / disc_93_6_4.pth load strictly), same sorted-key execution order
(pooling BEFORE batch-norm, AE_model.py:49) and the in-place ``size_list.reverse()`` of ``Decoder.forward`` (:166).

Every 5-D op runs on the HIP kernels: the three separable 1-D convs per block are HBM-bound stencils
(conv_generic.hip), MaxPool writes a 1-byte arg-max, and BatchNorm3d + LeakyReLU/ReLU run as ONE fused pass.
"""
import torch.nn as tnn

from ... import nn as mnn
from ... import ops


def _axis_conv(axis, cin, cout, k, s, p):
    ks, st, pd = [1, 1, 1], [1, 1, 1], [0, 0, 0]
    ks[axis], st[axis], pd[axis] = k, s, p
    return mnn.Conv3d(in_channels=cin, out_channels=cout, kernel_size=tuple(ks), stride=tuple(st), padding=tuple(pd))


def _separable(first_key_index, cin, cout, kw):
    names = ("convx", "convy", "convz")
    return {"%d_%s" % (first_key_index + a, names[a]): _axis_conv(a, cin if a == 0 else cout, cout, kw["conv_k"],
                                                                  kw["conv_s"], kw["conv_pad"]) for a in range(3)}


def _activation(kind):
    if kind == "l_relu":
        return mnn.LeakyReLU(), tnn.init.calculate_gain("leaky_relu", 0.01)
    return mnn.ReLU(), tnn.init.calculate_gain("relu")


class _InitMixin:
    def _init_weights(self, moddict):
        for _, m in moddict.items():
            if hasattr(m, "weight") and m.weight.dim() > 1:
                tnn.init.xavier_uniform_(m.weight.data, gain=self.init_gain)
                tnn.init.constant_(m.bias.data, 0)


class DownBlock(tnn.Module, _InitMixin):
    def __init__(self, c_in, c_out, skip=False, **kwargs):
        super().__init__()
        self.skip = skip
        layers = _separable(1, c_in, c_out, kwargs)
        layers["4_pooling"] = mnn.MaxPool3d(kernel_size=kwargs["maxpool_k"], stride=kwargs["maxpool_s"])
        self.block = tnn.ModuleDict(layers)
        if kwargs["batch_norm"]:
            self.block.update({"5_batch_norm": mnn.BatchNorm3d(c_out)})
        act, self.init_gain = _activation(kwargs["act"])
        self.block.update({"6_act": act})
        self.init_weights()

    def init_weights(self):
        self._init_weights(self.block)

    def forward(self, x):
        _, _, D, H, W = x.shape
        shape_before_pool = (D, H, W)
        x = mnn.run_fused([m for _, m in sorted(self.block.items())], x)
        return x, shape_before_pool


class UpBlock(tnn.Module, _InitMixin):
    def __init__(self, c_in, c_out, skip=False, **kwargs):
        super().__init__()
        self.skip = skip
        self.block = tnn.ModuleDict()
        if kwargs["up"] == "transpose_conv":
            self.block.update({"1_upsample": mnn.ConvTranspose3d(in_channels=c_in, out_channels=c_out,
                                                                 kernel_size=kwargs["scale"], stride=kwargs["scale"],
                                                                 padding=kwargs["t_conv_pad"])})
        else:
            self.block.update({"1_upsample": mnn.Upsample(scale_factor=kwargs["scale"], mode=kwargs["scale_mode"])})
        self.block.update(_separable(2, c_in, c_out, kwargs))
        if kwargs["batch_norm"]:
            self.block.update({"5_batch_norm": mnn.BatchNorm3d(c_out)})
        act, self.init_gain = _activation(kwargs["act"])
        self.block.update({"6_act": act})
        self.init_weights()

    def init_weights(self):
        self._init_weights(self.block)

    def forward(self, x, shape_before_pool=None, x_before_pool=None):
        mods = [m for _, m in sorted(self.block.items())]
        up, conv = mods[0], mods[1]   # "1_upsample", "2_convx"
        if (isinstance(up, mnn.Upsample) and up.mode == "nearest" and up.size is None and isinstance(up.scale_factor, (int, float))
                and all(shape_before_pool[a] <= int(x.shape[2 + a] * up.scale_factor) for a in range(3))
                and ops.upsample_conv3d_supported(x, conv.weight, up.scale_factor, conv.stride, conv.padding, conv.dilation)):
            # the upsampled tensor is never formed (the last block: 8 channels at 160x192x160): the first convolution reads the
            # coarse tensor through the nearest-neighbour index map (ops.upsample_conv3d)
            x = ops.upsample_conv3d(x, int(up.scale_factor), conv.weight, conv.bias, conv.padding)
            return mnn.run_fused(mods[2:], x)
        x = up(x)
        if any(shape_before_pool[a] > x.shape[2 + a] for a in range(3)):  # odd sizes: nearest resize to the skip size
            x = ops.upsample3d(x, size=tuple(shape_before_pool[:3]), mode="nearest")
        return mnn.run_fused(mods[1:], x)


class Encoder(tnn.Module):
    def __init__(self, **kwargs):
        super().__init__()
        self.encode = tnn.ModuleList()
        if kwargs["reduce_size"]:
            self.encode.append(mnn.Conv3d(1, 1, kernel_size=4, stride=4, padding=0))
        for i in range(kwargs["deapth"]):
            self.encode.append(DownBlock(c_in=kwargs["chanels"][i], c_out=kwargs["chanels"][i + 1],
                                         skip=kwargs["skip_map"][i], **kwargs["down_block_kwargs"]))

    def forward(self, x):
        size_list = []
        for module in self.encode:
            x, size = module(x)  # (the reference's reduce_size conv returns a bare tensor here and fails the same way)
            size_list.append(size)
        return x, size_list


class Decoder(tnn.Module):
    def __init__(self, **kwargs):
        super().__init__()
        self.decode = tnn.ModuleList()
        for i in range(kwargs["deapth"]):
            self.decode.append(UpBlock(c_in=kwargs["chanels"][i], c_out=kwargs["chanels"][i + 1],
                                       skip=kwargs["skip_map"][i], **kwargs["up_block_kwargs"]))
        if kwargs["reduce_size"]:
            self.decode.append(mnn.ConvTranspose3d(1, 1, kernel_size=4, stride=4, padding=0))
        self.vox = mnn.Conv3d(in_channels=1, out_channels=1, kernel_size=3, stride=1, padding=1)

    def forward(self, x, size_list):
        size_list.reverse()
        for i, module in enumerate(self.decode):
            x = module(x, size_list[i])
        return self.vox(x)


class AE(tnn.Module):
    def __init__(self, **kwargs):
        super().__init__()
        if kwargs["is_skip"]:
            skip_map = kwargs["skip_map"]
            assert len(skip_map) < kwargs["deapth"], "skip map len shold mutch deapth"
        else:
            skip_map = [False for _ in range(kwargs["deapth"])]
        chanels = [kwargs["c_in"]]
        c = kwargs["c_base"]
        for _ in range(kwargs["deapth"]):
            chanels.append(c)
            c = kwargs["inc_size"] * c
        self.enc = Encoder(deapth=kwargs["deapth"], chanels=chanels, skip_map=skip_map,
                           reduce_size=kwargs["reduce_size"], down_block_kwargs=kwargs["down_block_kwargs"])
        self.dec = Decoder(deapth=kwargs["deapth"], chanels=chanels[::-1], skip_map=skip_map[::-1],
                           reduce_size=kwargs["reduce_size"], up_block_kwargs=kwargs["up_block_kwargs"])

    def forward(self, x):
        x, size_list = self.enc(x)
        return self.dec(x, size_list)


class _LatentHead(tnn.Module, _InitMixin):
    """Shared body of Discriminator / Classificator: 3 separable convs on the latent, then a tiny MLP (torch ops:
    the (N, 64) vectors are not on the volumetric path)."""
    _attr = None
    _out_key = None

    def __init__(self, **kwargs):
        super().__init__()
        layers = _separable(1, kwargs["c_in"], kwargs["c_out"], kwargs)
        layers["4_flat"] = mnn.Flatten()
        layers["5_l1"] = tnn.Linear(kwargs["l_in"], kwargs["l_out"])
        md = tnn.ModuleDict(layers)
        if kwargs["batch_norm"]:
            md.update({"6_batch_norm": tnn.BatchNorm1d(kwargs["l_out"])})
        act, self.init_gain = _activation(kwargs["act"])
        md.update({"7_act": act})
        md.update({"8_drop": tnn.Dropout(kwargs["p_drop"])})
        md.update({"9_l_f": tnn.Linear(kwargs["l_out"], kwargs[self._out_key])})
        setattr(self, self._attr, md)
        self.init_weights()

    def init_weights(self):
        self._init_weights(getattr(self, self._attr))

    def forward(self, x):
        return mnn.run_fused([m for _, m in sorted(getattr(self, self._attr).items())], x)


class Discriminator(_LatentHead):
    _attr = "disc"
    _out_key = "n_domains"


class Classificator(_LatentHead):
    _attr = "clf"
    _out_key = "n_class"
