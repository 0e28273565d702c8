"""CPU restatement of classification/models/AE_model.py (test oracle): separable (k,1,1)/(1,k,1)/(1,1,k) conv
Down/Up blocks, Encoder/Decoder/AE, Discriminator, Classificator.  Same ModuleDict keys (state_dict compatible),
same construction order (so a seeded construction reproduces the reference's parameters bit-for-bit), and the
reference's sorted-key execution order: pooling BEFORE batch-norm (AE_model.py:49), Decoder reverses the caller's
size list in place (AE_model.py:166).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

_AXIS_KEYS = ("convx", "convy", "convz")


def _sep_convs(first_index, cin, cout, k, s, p):
    out = {}
    for a, name in enumerate(_AXIS_KEYS):
        ks, st, pd = [1, 1, 1], [1, 1, 1], [0, 0, 0]
        ks[a], st[a], pd[a] = k, s, p
        out["%d_%s" % (first_index + a, name)] = nn.Conv3d(cin if a == 0 else cout, cout, tuple(ks), tuple(st), tuple(pd))
    return out


def _act_and_gain(kind):
    if kind == "l_relu":
        return nn.LeakyReLU(), nn.init.calculate_gain("leaky_relu", 0.01)
    return nn.ReLU(), nn.init.calculate_gain("relu")


def _xavier(moddict, gain):
    for m in moddict.values():
        if hasattr(m, "weight") and m.weight.dim() > 1:
            nn.init.xavier_uniform_(m.weight.data, gain=gain)
            nn.init.constant_(m.bias.data, 0)


class DownBlock(nn.Module):  # AE_model.py:4-53
    def __init__(self, c_in, c_out, skip=False, **kw):
        super().__init__()
        self.skip = skip
        d = _sep_convs(1, c_in, c_out, kw["conv_k"], kw["conv_s"], kw["conv_pad"])
        d["4_pooling"] = nn.MaxPool3d(kernel_size=kw["maxpool_k"], stride=kw["maxpool_s"])
        self.block = nn.ModuleDict(d)
        if kw["batch_norm"]:
            self.block.update({"5_batch_norm": nn.BatchNorm3d(c_out)})
        act, self.init_gain = _act_and_gain(kw["act"])
        self.block.update({"6_act": act})
        _xavier(self.block, self.init_gain)

    def forward(self, x):
        shape = tuple(x.shape[2:])
        for _, m in sorted(self.block.items()):
            x = m(x)
        return x, shape


class UpBlock(nn.Module):  # AE_model.py:56-120
    def __init__(self, c_in, c_out, skip=False, **kw):
        super().__init__()
        self.skip = skip
        self.block = nn.ModuleDict()
        if kw["up"] == "transpose_conv":
            self.block.update({"1_upsample": nn.ConvTranspose3d(c_in, c_out, kw["scale"], kw["scale"], kw["t_conv_pad"])})
        else:
            self.block.update({"1_upsample": nn.Upsample(scale_factor=kw["scale"], mode=kw["scale_mode"])})
        self.block.update(_sep_convs(2, c_in, c_out, kw["conv_k"], kw["conv_s"], kw["conv_pad"]))
        if kw["batch_norm"]:
            self.block.update({"5_batch_norm": nn.BatchNorm3d(c_out)})
        act, self.init_gain = _act_and_gain(kw["act"])
        self.block.update({"6_act": act})
        _xavier(self.block, self.init_gain)

    def forward(self, x, shape_before_pool=None, x_before_pool=None):
        for key, m in sorted(self.block.items()):
            x = m(x)
            if key == "1_upsample" and any(shape_before_pool[a] > x.shape[2 + a] for a in range(3)):
                x = F.interpolate(x, tuple(shape_before_pool))
        return x


class Encoder(nn.Module):  # AE_model.py:123-144
    def __init__(self, **kw):
        super().__init__()
        self.encode = nn.ModuleList()
        if kw["reduce_size"]:
            self.encode.append(nn.Conv3d(1, 1, kernel_size=4, stride=4, padding=0))
        for i in range(kw["deapth"]):
            self.encode.append(DownBlock(kw["chanels"][i], kw["chanels"][i + 1], kw["skip_map"][i], **kw["down_block_kwargs"]))

    def forward(self, x):
        sizes = []
        for m in self.encode:
            x, s = m(x)
            sizes.append(s)
        return x, sizes


class Decoder(nn.Module):  # AE_model.py:147-170
    def __init__(self, **kw):
        super().__init__()
        self.decode = nn.ModuleList()
        for i in range(kw["deapth"]):
            self.decode.append(UpBlock(kw["chanels"][i], kw["chanels"][i + 1], kw["skip_map"][i], **kw["up_block_kwargs"]))
        if kw["reduce_size"]:
            self.decode.append(nn.ConvTranspose3d(1, 1, kernel_size=4, stride=4, padding=0))
        self.vox = nn.Conv3d(1, 1, kernel_size=3, stride=1, padding=1)

    def forward(self, x, size_list):
        size_list.reverse()
        for i, m in enumerate(self.decode):
            x = m(x, size_list[i])
        return self.vox(x)


class AE(nn.Module):  # AE_model.py:173-210
    def __init__(self, **kw):
        super().__init__()
        if kw["is_skip"]:
            skip_map = kw["skip_map"]
            assert len(skip_map) < kw["deapth"], "skip map len shold mutch deapth"
        else:
            skip_map = [False] * kw["deapth"]
        chans, c = [kw["c_in"]], kw["c_base"]
        for _ in range(kw["deapth"]):
            chans.append(c)
            c = kw["inc_size"] * c
        self.enc = Encoder(deapth=kw["deapth"], chanels=chans, skip_map=skip_map, reduce_size=kw["reduce_size"],
                           down_block_kwargs=kw["down_block_kwargs"])
        self.dec = Decoder(deapth=kw["deapth"], chanels=chans[::-1], skip_map=skip_map[::-1],
                           reduce_size=kw["reduce_size"], up_block_kwargs=kw["up_block_kwargs"])

    def forward(self, x):
        x, sizes = self.enc(x)
        return self.dec(x, sizes)


def _head(attr_name, final_key, n_out_key):
    class Head(nn.Module):  # AE_model.py:213-262 (Discriminator) / :264-312 (Classificator)
        def __init__(self, **kw):
            super().__init__()
            d = _sep_convs(1, kw["c_in"], kw["c_out"], kw["conv_k"], kw["conv_s"], kw["conv_pad"])
            d["4_flat"] = nn.Flatten()
            d["5_l1"] = nn.Linear(kw["l_in"], kw["l_out"])
            md = nn.ModuleDict(d)
            if kw["batch_norm"]:
                md.update({"6_batch_norm": nn.BatchNorm1d(kw["l_out"])})
            act, self.init_gain = _act_and_gain(kw["act"])
            md.update({"7_act": act})
            md.update({"8_drop": nn.Dropout(kw["p_drop"])})
            md.update({final_key: nn.Linear(kw["l_out"], kw[n_out_key])})
            setattr(self, attr_name, md)
            _xavier(md, self.init_gain)

        def forward(self, x):
            for _, m in sorted(getattr(self, attr_name).items()):
                x = m(x)
            return x

    return Head


Discriminator = _head("disc", "9_l_f", "n_domains")
Discriminator.__name__ = Discriminator.__qualname__ = "Discriminator"
Classificator = _head("clf", "9_l_f", "n_class")
Classificator.__name__ = Classificator.__qualname__ = "Classificator"
