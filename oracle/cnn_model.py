"""CPU restatement of classification/models/cnn_model.py (test oracle): CNN (:104-175), VoxResNet (:43-101) with
BasicBlock (:17-40), DilatedCNN (:207-256).  Module names inside ``self.model`` match the reference so state_dicts
interchange; layer tables are data-driven instead of the reference's unrolled add_module calls.

Quirk kept: VoxResNet registers "activation_6" twice (cnn_model.py:83 and :95); nn.Module.add_module keeps the
first position, so with n_blocks >= 4 there is no activation after fully_conn_1.
"""
import numpy as np
import torch.nn as nn


class Flatten(nn.Module):
    def forward(self, x):
        return x.view(x.size(0), -1)


class BasicBlock(nn.Module):
    def __init__(self, inplanes, planes, stride=1):
        super().__init__()
        self.conv1 = nn.Conv3d(inplanes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn1 = nn.BatchNorm3d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv3d(planes, planes, 3, padding=1, bias=False)
        self.bn2 = nn.BatchNorm3d(planes)
        self.stride = stride

    def forward(self, x):
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        out = out + x
        return self.relu(out)


class VoxResNet(nn.Module):
    def __init__(self, input_shape=(128, 128, 128), num_classes=2, n_filters=32, stride=2, n_blocks=3,
                 n_flatten_units=None, dropout=0, n_fc_units=128):
        super().__init__()
        f = n_filters
        m = nn.Sequential()
        m.add_module("conv3d_1", nn.Conv3d(1, f, 3, padding=1, stride=stride))
        m.add_module("batch_norm_1", nn.BatchNorm3d(f))
        m.add_module("activation_1", nn.ReLU(inplace=True))
        m.add_module("conv3d_2", nn.Conv3d(f, f, 3, padding=1))
        m.add_module("batch_norm_2", nn.BatchNorm3d(f))
        m.add_module("activation_2", nn.ReLU(inplace=True))
        stages = [(f, 2 * f), (2 * f, 2 * f), (2 * f, 4 * f), (4 * f, 4 * f)]
        for s, (cin, cout) in enumerate(stages[:max(n_blocks, 1)], start=1):
            if s > 1 and n_blocks < s:
                break
            m.add_module("conv3d_%d" % (s + 2), nn.Conv3d(cin, cout, 3, padding=1, stride=2))
            m.add_module("block_%d" % (2 * s - 1), BasicBlock(cout, cout))
            m.add_module("block_%d" % (2 * s), BasicBlock(cout, cout))
            m.add_module("batch_norm_%d" % (s + 2), nn.BatchNorm3d(cout))
            m.add_module("activation_%d" % (s + 2), nn.ReLU(inplace=True))
        if n_flatten_units is None:
            n_flatten_units = 4 * f * np.prod(np.array(input_shape) // (2 ** n_blocks * stride))
        m.add_module("flatten_1", Flatten())
        m.add_module("fully_conn_1", nn.Linear(int(n_flatten_units), n_fc_units))
        m.add_module("activation_6", nn.ReLU(inplace=True))
        m.add_module("dropout_1", nn.Dropout(dropout))
        m.add_module("fully_conn_2", nn.Linear(n_fc_units, num_classes))
        self.model = m

    def forward(self, x):
        return self.model(x)


class CNN(nn.Module):
    def __init__(self, input_shape=(64, 76, 48), n_filters=16, n_blocks=3, stride=1, n_fc_units=128):
        super().__init__()
        m = nn.Sequential()
        cin, idx = 1, 1
        for b in range(1, n_blocks + 1):
            cout = n_filters * 2 ** (b - 1)
            for j in range(2):
                st = stride if (b == 1 and j == 0) else 1
                m.add_module("conv3d_%d" % idx, nn.Conv3d(cin, cout, kernel_size=3, stride=st, padding=1))
                m.add_module("batch_norm_%d" % idx, nn.BatchNorm3d(cout))
                m.add_module("activation_%d" % idx, nn.ReLU(inplace=True))
                cin = cout
                idx += 1
            m.add_module("max_pool3d_%d" % b, nn.MaxPool3d(kernel_size=2))
        m.add_module("flatten_1", Flatten())
        div = 2 ** n_blocks * stride
        feat = cin * (input_shape[0] // div) * (input_shape[1] // div) * (input_shape[2] // div)
        m.add_module("fully_conn_1", nn.Linear(feat, n_fc_units))
        m.add_module("batch_norm_9", nn.BatchNorm1d(n_fc_units))
        m.add_module("activation_9", nn.ReLU(inplace=True))
        self.model = m

    def forward(self, x):
        return self.model(x)


class DilatedCNN(nn.Module):
    def __init__(self, input_shape=(180, 180, 180), n_channels=32):
        super().__init__()
        c = n_channels
        # (cin, cout, stride, padding) of the six dilation-3 convs, and where the two MaxPool3d(4,2) sit
        spec = [(1, c, 2, 0), (c, c, 1, 3), (c, 2 * c, 2, 0), (2 * c, 2 * c, 1, 3), (2 * c, 4 * c, 1, 3), (4 * c, 4 * c, 1, 0)]
        m = nn.Sequential()
        for i, (cin, cout, st, pad) in enumerate(spec, start=1):
            m.add_module("conv3d_%d" % i, nn.Conv3d(cin, cout, kernel_size=3, stride=st, dilation=3, padding=pad))
            m.add_module("batch_norm_%d" % i, nn.BatchNorm3d(cout))
            m.add_module("activation_%d" % i, nn.LeakyReLU())
            if i in (2, 4):
                m.add_module("max_pool3d_%d" % (i // 2), nn.MaxPool3d(kernel_size=4, stride=2))
        m.add_module("flatten_1", Flatten())
        m.add_module("fully_conn_1", nn.Linear(4 * c * ((input_shape[0] - 61) // 16 - 5) ** 3, 256))
        m.add_module("activation_7", nn.LeakyReLU())
        m.add_module("fully_conn_2", nn.Linear(256, 128))
        m.add_module("activation_8", nn.LeakyReLU())
        m.add_module("fully_conn_3", nn.Linear(128, 2))
        m.add_module("softmax", nn.Softmax(dim=-1))
        self.model = m

    def forward(self, x):
        return self.model(x)
