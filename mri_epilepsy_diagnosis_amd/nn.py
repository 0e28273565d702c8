"""Drop-in ``torch.nn`` layer classes whose forward/backward run on the HIP kernels.

Each class subclasses its ``torch.nn`` namesake, so constructor signatures, parameter initialisation and
``state_dict`` keys are identical to what the reference models create; only ``forward`` is replaced.
``fused_norm_act`` applies a normalisation layer and the activation that follows it in ONE kernel pass — the
conv -> BatchNorm3d -> PReLU blocks of ``unet.UNet`` and the (MaxPool ->) BatchNorm3d -> LeakyReLU tails of
``AE_model.DownBlock`` (classification/models/AE_model.py:27-36) — and ``FusedSequential`` does that pairing
automatically for ``nn.Sequential``-style reference models (classification/models/cnn_model.py:104-175).
"""
import torch
import torch.nn as tnn

from . import ops


class Conv3d(tnn.Conv3d):
    def forward(self, x):
        if self.padding_mode != "zeros" or self.groups != 1 or isinstance(self.padding, str):
            raise NotImplementedError("mri3d Conv3d supports padding_mode='zeros', groups=1, numeric padding")
        return ops.conv3d(x, self.weight, self.bias, self.stride, self.padding, self.dilation)


class ConvTranspose3d(tnn.ConvTranspose3d):
    def forward(self, x, output_size=None):
        if output_size is not None or self.groups != 1 or self.padding_mode != "zeros":
            raise NotImplementedError("mri3d ConvTranspose3d supports groups=1, zero padding, no output_size")
        return ops.conv_transpose3d(x, self.weight, self.bias, self.stride, self.padding, self.output_padding,
                                    self.dilation)


def _act_spec(act):
    """(kind, alpha tensor or None, slope) for an activation module (or None)."""
    if act is None or isinstance(act, tnn.Identity):
        return None, None, 0.0
    if isinstance(act, tnn.PReLU):
        return "prelu", act.weight, 0.0
    if isinstance(act, tnn.LeakyReLU):
        return "leaky_relu", None, float(act.negative_slope)
    if isinstance(act, tnn.ReLU):
        return "relu", None, 0.0
    raise NotImplementedError("unsupported activation module %r" % (act,))


def fused_norm_act(norm, act, x, out=None):
    """act(norm(x)) in one pass.  `norm` is a BatchNorm3d / InstanceNorm3d module or None; `act` an activation or None.
    `out=(buffer, channel_offset)` makes the kernel write straight into a channel slice of a wider NDHWC buffer (a
    decoder concat buffer) and returns that slice as a view."""
    kind, alpha, slope = _act_spec(act)
    if norm is None:
        if kind is None and out is None:
            return x
        return ops.norm_act(x, None, None, alpha, None, None, "none", 0.1, 0.0, kind, slope, out)
    if isinstance(norm, tnn.modules.batchnorm._BatchNorm):
        use_batch = norm.training or norm.running_mean is None
        if norm.training and norm.track_running_stats and norm.num_batches_tracked is not None:
            norm.num_batches_tracked.add_(1)
        return ops.norm_act(x, norm.weight, norm.bias, alpha,
                            norm.running_mean if norm.track_running_stats else None,
                            norm.running_var if norm.track_running_stats else None,
                            ("sync" if (norm.training and ops.sync_batchnorm_reducer() is not None) else "batch")
                            if use_batch else "running",
                            norm.momentum, norm.eps, kind, slope, out)
    if isinstance(norm, tnn.modules.instancenorm._InstanceNorm):
        if norm.track_running_stats:
            raise NotImplementedError("InstanceNorm3d(track_running_stats=True) is not supported")
        return ops.norm_act(x, norm.weight, norm.bias, alpha, None, None, "instance", 0.1, norm.eps, kind, slope, out)
    if isinstance(norm, tnn.GroupNorm):
        return ops.norm_act(x, norm.weight, norm.bias, alpha, None, None, "group", 0.1, norm.eps, kind, slope, out,
                            norm.num_channels // norm.num_groups)
    raise NotImplementedError("unsupported normalisation module %r" % (norm,))


def conv_norm_act(conv, norm, act, x, out=None):
    """act(norm(conv(x))) for a Conv3d module followed by a normalisation (`unet.UNet`'s ConvolutionalBlock, the conv -> BN ->
    ReLU stems of cnn_model.py).  When `norm` is a BatchNorm3d that will use BATCH statistics (training mode, local statistics)
    the convolution is asked to accumulate them in its epilogue, which saves the statistics pass over its output."""
    wants = (isinstance(norm, tnn.modules.batchnorm._BatchNorm) and (norm.training or norm.running_mean is None)
             and ops.sync_batchnorm_reducer() is None and isinstance(conv, Conv3d) and conv.padding_mode == "zeros"
             and conv.groups == 1 and not isinstance(conv.padding, str))
    if isinstance(x, (tuple, list)):
        # x = (xa, xb): the convolution of torch.cat((xa, xb), dim=1), read from the two tensors (ops.conv3d_cat)
        xa, xb = x
        plain = (isinstance(conv, Conv3d) and conv.padding_mode == "zeros" and conv.groups == 1 and not isinstance(conv.padding, str)
                 and tuple(conv.stride) == (1, 1, 1) and tuple(conv.dilation) == (1, 1, 1))
        if plain:
            y = ops.conv3d_cat(xa, xb, conv.weight, conv.bias, conv.padding, bn_stats=wants)
        else:
            y = conv(ops.cat_channels([xa, xb]))
    elif wants:
        y = ops.conv3d(x, conv.weight, conv.bias, conv.stride, conv.padding, conv.dilation, bn_stats=True)
    else:
        y = conv(x)
    return fused_norm_act(norm, act, y, out)


class GroupNorm(tnn.GroupNorm):
    def forward(self, x):
        return fused_norm_act(self, None, x)


class BatchNorm3d(tnn.BatchNorm3d):
    def forward(self, x):
        return fused_norm_act(self, None, x)


class InstanceNorm3d(tnn.InstanceNorm3d):
    def forward(self, x):
        return fused_norm_act(self, None, x)


class PReLU(tnn.PReLU):
    def forward(self, x):
        if x.dim() != 5:  # classifier-head vectors (N, F): not on the volumetric path
            return super().forward(x)
        return fused_norm_act(None, self, x)


class ReLU(tnn.ReLU):
    def forward(self, x):
        if x.dim() != 5:  # classifier-head vectors (N, F): not on the volumetric path
            return super().forward(x)
        return fused_norm_act(None, self, x)


class LeakyReLU(tnn.LeakyReLU):
    def forward(self, x):
        if x.dim() != 5:  # classifier-head vectors (N, F): not on the volumetric path
            return super().forward(x)
        return fused_norm_act(None, self, x)


class MaxPool3d(tnn.MaxPool3d):
    def forward(self, x):
        if self.ceil_mode or self.return_indices or self.dilation not in (1, (1, 1, 1)):
            raise NotImplementedError("mri3d MaxPool3d supports floor mode, dilation 1, no indices")
        return ops.max_pool3d(x, self.kernel_size, self.stride, self.padding)

    def forward_with_skip(self, x):
        """(pool(x), x) as one autograd node — for blocks whose output also feeds a skip connection (ops.max_pool3d_skip)."""
        if self.ceil_mode or self.return_indices or self.dilation not in (1, (1, 1, 1)):
            raise NotImplementedError("mri3d MaxPool3d supports floor mode, dilation 1, no indices")
        return ops.max_pool3d_skip(x, self.kernel_size, self.stride, self.padding)


class Upsample(tnn.Upsample):
    def forward(self, x):
        return ops.upsample3d(x, self.size, self.scale_factor, self.mode, self.align_corners)


class Dropout3d(tnn.Dropout3d):
    def forward(self, x):
        return ops.dropout3d(x, self.p, self.training)


class Flatten(tnn.Module):
    """(N, C, D, H, W) -> (N, C*D*H*W) in torch's NCDHW element order (tiny tensors; a layout copy only)."""

    def forward(self, x):
        return x.contiguous(memory_format=torch.contiguous_format).view(x.size(0), -1)


_NORMS = (tnn.modules.batchnorm._BatchNorm, tnn.modules.instancenorm._InstanceNorm)
_ACTS = (tnn.PReLU, tnn.ReLU, tnn.LeakyReLU)


def run_fused(modules, x):
    """Run a list of modules like nn.Sequential, fusing 5-D [norm ->] activation pairs into one kernel."""
    mods = list(modules)
    i = 0
    while i < len(mods):
        m = mods[i]
        five_d = torch.is_tensor(x) and x.dim() == 5
        if (five_d and i == 0 and isinstance(m, Conv3d) and len(mods) > 2 and isinstance(mods[1], Conv3d) and isinstance(mods[2], Conv3d)
                and ops.conv3d_pair_supported(x, m, mods[1])):
            # the head of the autoencoder's first DownBlock on the network input: the gradient of the (k,1,1) convolution's output is
            # never formed (ops.conv3d_pair)
            x = ops.conv3d_pair(x, m, mods[1])
            i += 2
            continue
        if five_d and isinstance(m, Conv3d) and i + 1 < len(mods) and isinstance(mods[i + 1], tnn.BatchNorm3d):
            nxt2 = mods[i + 2] if i + 2 < len(mods) else None
            if isinstance(nxt2, _ACTS):
                x = conv_norm_act(m, mods[i + 1], nxt2, x)
                i += 3
            else:
                x = conv_norm_act(m, mods[i + 1], None, x)
                i += 2
            continue
        if five_d and isinstance(m, (tnn.BatchNorm3d, tnn.InstanceNorm3d, tnn.GroupNorm)):
            nxt = mods[i + 1] if i + 1 < len(mods) else None
            if isinstance(nxt, _ACTS):
                x = fused_norm_act(m, nxt, x)
                i += 2
                continue
            x = fused_norm_act(m, None, x)
        elif five_d and isinstance(m, _ACTS):
            x = fused_norm_act(None, m, x)
        else:
            x = m(x)
        i += 1
    return x


class FusedSequential(tnn.Sequential):
    def forward(self, x):
        return run_fused(self, x)
