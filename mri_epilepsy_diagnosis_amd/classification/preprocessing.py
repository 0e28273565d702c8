"""Histogram standardisation of a T1w volume on the device — drop-in for `normalize(tensor, landmarks, mask, cutoff,
epsilon)` of the reference's collate function (classification/train_ENC_CLF.ipynb cell 9, a restatement of TorchIO's
HistogramStandardization; landmarks file segmentation/weights/fcd_train_data_landmarks.npy).

The two passes over the ~5-7 M voxels run as HIP kernels (`mri3d_order_stats_f32`: exact order statistics by radix
histograms; `mri3d_piecewise_linear_f32`: the float64 landmark map); the 13-number bookkeeping in between (percentile
interpolation, slopes, intercepts) is the reference's own float64 numpy arithmetic, so the result is bit-identical to the
reference for the same input.  SURVEY §8f row 2.
"""
import ctypes

import numpy as np
import torch

from .. import _lib
from ..ops import _ptr, _stream, _workspace

DEFAULT_CUTOFF = 0.01, 0.99
STANDARD_RANGE = 0, 100
# landmark / percentile indices the reference uses (it drops the 25th and 75th percentile landmarks)
RANGE_TO_USE = [0, 1, 2, 4, 5, 6, 7, 8, 10, 11, 12]


def _standardize_cutoff(cutoff):
    cutoff = np.asarray(cutoff, dtype=np.float64).copy()
    cutoff[0] = max(0., cutoff[0])
    cutoff[1] = min(1., cutoff[1])
    cutoff[0] = np.min([cutoff[0], 0.09])
    cutoff[1] = np.max([cutoff[1], 0.91])
    return cutoff


def _get_percentiles(percentiles_cutoff):
    quartiles = np.arange(25, 100, 25).tolist()
    deciles = np.arange(10, 100, 10).tolist()
    return np.array(sorted(set(list(percentiles_cutoff) + quartiles + deciles)))


def order_statistics(x, ranks):
    """x_(r) for every r in `ranks` (0-based ranks into the ascending order of the flattened fp32 device tensor)."""
    if not x.is_cuda or x.dtype != torch.float32:
        raise RuntimeError("order_statistics: needs a float32 ROCm device tensor (got %s on %s); there is no CPU fallback"
                           % (x.dtype, x.device))
    L = _lib.lib()
    x = x.contiguous().view(-1)
    ranks = np.ascontiguousarray(ranks, dtype=np.int64)
    out = torch.empty(len(ranks), dtype=torch.float32, device=x.device)
    ws = _workspace(L.mri3d_order_stats_workspace_bytes(), x.device)
    _lib.check(L.mri3d_order_stats_f32(_ptr(x), x.numel(), ranks.ctypes.data_as(ctypes.c_void_p), len(ranks), _ptr(out),
                                       _ptr(ws), ws.numel(), _stream()), "order_stats")
    return out


def percentile(x, percentiles):
    """np.percentile(x, percentiles) (method 'linear') of a float32 device tensor, bit-identical to numpy 2.x: the two
    neighbouring order statistics come from the device, the interpolation is numpy's `_lerp` arithmetic in float64."""
    n = x.numel()
    quantiles = np.true_divide(np.asarray(percentiles, dtype=np.float64), 100)
    virtual = (n - 1) * quantiles
    prev = np.floor(virtual)
    nxt = prev + 1
    above = virtual >= n - 1
    prev[above] = n - 1
    nxt[above] = n - 1
    below = virtual < 0
    prev[below] = 0
    nxt[below] = 0
    prev_i, nxt_i = prev.astype(np.intp), nxt.astype(np.intp)
    gamma = np.asanyarray(virtual - np.floor(virtual), dtype=virtual.dtype)
    vals = order_statistics(x, np.concatenate([prev_i, nxt_i])).cpu().numpy()     # 2 x 13 floats cross PCIe
    a, b = vals[:len(prev_i)], vals[len(prev_i):]
    diff_b_a = np.subtract(b, a)                                 # float32, as in numpy's _lerp
    lerp = np.asanyarray(np.add(a, diff_b_a * gamma))            # float64
    np.subtract(b, diff_b_a * (1 - gamma), out=lerp, where=gamma >= 0.5, casting="unsafe", dtype=type(lerp.dtype))
    return lerp


def normalize(tensor, landmarks, mask=None, cutoff=None, epsilon=1e-5):
    """Reference `normalize` (train_ENC_CLF.ipynb cell 9) for a float32 volume that already lives on the device."""
    if not tensor.is_cuda:
        raise RuntimeError("normalize runs only on a ROCm device tensor (got %s); there is no CPU fallback" % tensor.device)
    cutoff_ = DEFAULT_CUTOFF if cutoff is None else cutoff
    mapping = np.asarray(landmarks)
    shape = tensor.shape
    data = tensor.reshape(-1).to(torch.float32).contiguous()
    sel = data if mask is None else data[torch.as_tensor(mask, device=data.device).reshape(-1).bool()]

    quantiles_cutoff = _standardize_cutoff(cutoff_)
    percentiles_cutoff = 100 * np.array(quantiles_cutoff)
    percentiles = _get_percentiles(percentiles_cutoff)
    percentile_values = percentile(sel, percentiles)

    range_mapping = mapping[RANGE_TO_USE]
    range_perc = percentile_values[RANGE_TO_USE]
    diff_mapping = np.diff(range_mapping)
    diff_perc = np.diff(range_perc)
    diff_perc[diff_perc < epsilon] = np.inf        # two equal landmarks in this image: flat segment
    slope = diff_mapping / diff_perc
    intercept = range_mapping[:-1] - slope * range_perc[:-1]
    edges = np.ascontiguousarray(range_perc[1:-1], dtype=np.float64)
    slope = np.ascontiguousarray(slope, dtype=np.float64)
    intercept = np.ascontiguousarray(intercept, dtype=np.float64)

    L = _lib.lib()
    out = torch.empty_like(data)
    as_p = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    _lib.check(L.mri3d_piecewise_linear_f32(_ptr(data), _ptr(out), data.numel(), as_p(edges), as_p(slope), as_p(intercept),
                                            len(slope), _stream()), "piecewise_linear")
    return out.reshape(shape)


def default_collate(batch, landmarks, device="cuda"):
    """collate_fn of the reference (cell 9) with the volumes standardised on the device: batch = [(X, y, domain), ...]."""
    X = torch.stack([normalize(item[0].to(device), landmarks) for item in batch])
    y = torch.as_tensor([item[1] for item in batch], dtype=torch.long)
    domain = torch.as_tensor([item[2] for item in batch], dtype=torch.long)
    return X, y, domain


def z_normalize(tensor):
    """TorchIO ZNormalization(masking_method=mean): (x - mean(x[x > mean(x)])) / std(x[x > mean(x)]) (unbiased std), whole
    tensor at once, on the device.  Returns (normalised tensor, stats) with stats = float64 tensor
    [mean of all voxels, masked count, masked mean, masked std]."""
    if not tensor.is_cuda or tensor.dtype != torch.float32:
        raise RuntimeError("z_normalize: needs a float32 ROCm device tensor (got %s on %s); there is no CPU fallback"
                           % (tensor.dtype, tensor.device))
    L = _lib.lib()
    x = tensor.contiguous()
    y = torch.empty_like(x)
    stats = torch.empty(4, dtype=torch.float64, device=x.device)
    ws = _workspace(L.mri3d_znorm_workspace_bytes(), x.device)
    _lib.check(L.mri3d_znorm_mean_mask_f32(_ptr(x), _ptr(y), x.numel(), _ptr(stats), _ptr(ws), ws.numel(), _stream()), "znorm")
    return y, stats


def crop_or_pad(tensor, target_shape, fill=0.0):
    """TorchIO CropOrPad(target_shape): centred crop / zero-pad of the last three axes of a (..., D, H, W) device tensor."""
    if not tensor.is_cuda or tensor.dtype != torch.float32:
        raise RuntimeError("crop_or_pad: needs a float32 ROCm device tensor (got %s on %s); there is no CPU fallback"
                           % (tensor.dtype, tensor.device))
    L = _lib.lib()
    x = tensor.contiguous()
    d, h, w = (int(v) for v in x.shape[-3:])
    td, th, tw = (int(v) for v in target_shape)
    outer = x.numel() // (d * h * w)
    y = torch.empty(tuple(x.shape[:-3]) + (td, th, tw), dtype=x.dtype, device=x.device)
    _lib.check(L.mri3d_crop_or_pad_f32(_ptr(x), _ptr(y), outer, d, h, w, td, th, tw, float(fill), _stream()), "crop_or_pad")
    return y
