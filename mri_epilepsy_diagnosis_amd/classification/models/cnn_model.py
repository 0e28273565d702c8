"""MI355X-native drop-in for classification/models/cnn_model.py: CNN (config-5 stand-in, cnn_model.py:104-175),
VoxResNet + BasicBlock (:17-101), DilatedCNN (:207-256), ConvLSTM (:178-204, CNN + torch LSTM).
Same constructor signatures and ``self.model`` child names, so reference state_dicts interchange.  The Sequential
containers are ``FusedSequential``: every BatchNorm3d -> (Leaky)ReLU pair is one fused HIP pass, convs/pools are HIP
kernels; Flatten/Linear/BatchNorm1d/Dropout/Softmax on the (N, F) head stay torch ops.
"""
import numpy as np
import torch.nn as tnn

from ... import nn as mnn
from ... import ops


class Flatten(mnn.Flatten):
    pass


def conv3x3x3(in_planes, out_planes, stride=1):
    return mnn.Conv3d(in_planes, out_planes, kernel_size=3, stride=stride, padding=1, bias=False)


class BasicBlock(tnn.Module):
    def __init__(self, inplanes, planes, stride=1):
        super().__init__()
        self.conv1 = conv3x3x3(inplanes, planes, stride)
        self.bn1 = mnn.BatchNorm3d(planes)
        self.relu = mnn.ReLU(inplace=True)
        self.conv2 = conv3x3x3(planes, planes)
        self.bn2 = mnn.BatchNorm3d(planes)
        self.stride = stride

    def forward(self, x):
        out = mnn.fused_norm_act(self.bn1, self.relu, self.conv1(x))
        out = mnn.fused_norm_act(self.bn2, None, self.conv2(out))
        return self.relu(ops.add(out, x))


def _bn_relu(seq, idx, channels, leaky=False):
    seq.add_module("batch_norm_%d" % idx, mnn.BatchNorm3d(channels))
    seq.add_module("activation_%d" % idx, mnn.LeakyReLU() if leaky else mnn.ReLU(inplace=True))


class VoxResNet(tnn.Module):
    def __init__(self, input_shape=(128, 128, 128), num_classes=2, n_filters=32, stride=2, n_blocks=3,
                 n_flatten_units=None, dropout=0, n_fc_units=128):
        super().__init__()
        f = n_filters
        self.model = mnn.FusedSequential()
        self.model.add_module("conv3d_1", mnn.Conv3d(1, f, kernel_size=3, padding=1, stride=stride))
        _bn_relu(self.model, 1, f)
        self.model.add_module("conv3d_2", mnn.Conv3d(f, f, kernel_size=3, padding=1))
        _bn_relu(self.model, 2, f)
        widths = [(f, 2 * f), (2 * f, 2 * f), (2 * f, 4 * f), (4 * f, 4 * f)]
        for stage in range(1, 5):
            if stage > 1 and n_blocks < stage:
                continue
            cin, cout = widths[stage - 1]
            self.model.add_module("conv3d_%d" % (stage + 2), mnn.Conv3d(cin, cout, kernel_size=3, padding=1, stride=2))
            self.model.add_module("block_%d" % (2 * stage - 1), BasicBlock(cout, cout))
            self.model.add_module("block_%d" % (2 * stage), BasicBlock(cout, cout))
            _bn_relu(self.model, stage + 2, cout)
        if n_flatten_units is None:
            n_flatten_units = 4 * f * np.prod(np.array(input_shape) // (2 ** n_blocks * stride))
        self.model.add_module("flatten_1", Flatten())
        self.model.add_module("fully_conn_1", tnn.Linear(int(n_flatten_units), n_fc_units))
        # the reference re-registers "activation_6" here (cnn_model.py:95): with n_blocks >= 4 it keeps its old slot
        self.model.add_module("activation_6", mnn.ReLU(inplace=True))
        self.model.add_module("dropout_1", tnn.Dropout(dropout))
        self.model.add_module("fully_conn_2", tnn.Linear(n_fc_units, num_classes))

    def forward(self, x):
        return self.model(x)


class CNN(tnn.Module):
    def __init__(self, input_shape=(64, 76, 48), n_filters=16, n_blocks=3, stride=1, n_fc_units=128):
        super().__init__()
        self.model = mnn.FusedSequential()
        cin, idx = 1, 1
        for blk in range(1, n_blocks + 1):
            cout = n_filters * 2 ** (blk - 1)
            for j in range(2):
                first = blk == 1 and j == 0
                self.model.add_module("conv3d_%d" % idx, mnn.Conv3d(cin, cout, kernel_size=3,
                                                                    stride=stride if first else 1, padding=1))
                _bn_relu(self.model, idx, cout)
                cin, idx = cout, idx + 1
            self.model.add_module("max_pool3d_%d" % blk, mnn.MaxPool3d(kernel_size=2))
        self.model.add_module("flatten_1", Flatten())
        div = 2 ** n_blocks * stride
        self.model.add_module("fully_conn_1", tnn.Linear(
            cin * (input_shape[0] // div) * (input_shape[1] // div) * (input_shape[2] // div), n_fc_units))
        self.model.add_module("batch_norm_9", tnn.BatchNorm1d(n_fc_units))
        self.model.add_module("activation_9", mnn.ReLU(inplace=True))

    def forward(self, x):
        return self.model(x)


class ConvLSTM(tnn.Module):
    def __init__(self, input_shape=(48, 64, 32), n_outputs=1, hidden_size=128, n_layers=2, n_fc_units_rnn=128,
                 dropout=0, stride=1, n_filters=16, n_blocks=3, n_fc_units_cnn=128):
        super().__init__()
        self.model = CNN(input_shape, n_filters, n_blocks, stride, n_fc_units_cnn)
        self.hidden_size = hidden_size
        self.n_layers = n_layers
        self.lstm = tnn.LSTM(n_fc_units_cnn, hidden_size, n_layers, batch_first=True, dropout=dropout)
        self.fc1 = tnn.Linear(hidden_size, n_fc_units_rnn)
        self.relu = tnn.ReLU(inplace=True)
        self.fc2 = tnn.Linear(n_fc_units_rnn, n_outputs)

    def forward(self, x):
        n_objects, seq_length = x.size()[0:2]
        x = x.contiguous().view([n_objects * seq_length] + list(x.size()[2:]))
        x = self.model(x).contiguous().view([n_objects, seq_length, -1])
        out, _ = self.lstm(x)
        return self.fc2(self.relu(self.fc1(out[:, -1, :])))


class DilatedCNN(tnn.Module):
    def __init__(self, input_shape=(180, 180, 180), n_channels=32):
        super().__init__()
        c = n_channels
        table = [(1, c, 2, 0), (c, c, 1, 3), (c, 2 * c, 2, 0), (2 * c, 2 * c, 1, 3), (2 * c, 4 * c, 1, 3),
                 (4 * c, 4 * c, 1, 0)]  # (cin, cout, stride, padding); kernel 3, dilation 3 throughout
        self.model = mnn.FusedSequential()
        for i, (cin, cout, st, pad) in enumerate(table, start=1):
            self.model.add_module("conv3d_%d" % i, mnn.Conv3d(cin, cout, kernel_size=3, stride=st, dilation=3, padding=pad))
            _bn_relu(self.model, i, cout, leaky=True)
            if i in (2, 4):
                self.model.add_module("max_pool3d_%d" % (i // 2), mnn.MaxPool3d(kernel_size=4, stride=2))
        self.model.add_module("flatten_1", Flatten())
        self.model.add_module("fully_conn_1", tnn.Linear(4 * c * ((input_shape[0] - 61) // 16 - 5) ** 3, 256))
        self.model.add_module("activation_7", mnn.LeakyReLU())
        self.model.add_module("fully_conn_2", tnn.Linear(256, 128))
        self.model.add_module("activation_8", mnn.LeakyReLU())
        self.model.add_module("fully_conn_3", tnn.Linear(128, 2))
        self.model.add_module("softmax", tnn.Softmax(dim=-1))

    def forward(self, x):
        return self.model(x)
