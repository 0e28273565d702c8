"""Average surface distance of two binary masks on the device — what `calculate_metrics` asks of
compute_average_surface_distance(compute_surface_distances(surface, prediction, spacing_mm=(1,1,1)))
(segmentation/routine.py:205-214; segmentation/metrics.py:25-207).  SURVEY §8f row 4.

The surface-element area table (256 float64, unit spacing) in data/surfel_area_spacing111.npy is DATA derived by
oracle/gen_golden.py from the lookup table of the reference's metrics.py with the reference's own formula
(metrics.py:57-71); only unit spacing — the one the reference uses — is shipped.
"""
import os

import numpy as np
import torch

from .. import _lib
from ..ops import _ptr, _stream, _workspace
import ctypes

_AREA = None


def _area_table():
    global _AREA
    if _AREA is None:
        _AREA = np.ascontiguousarray(np.load(os.path.join(os.path.dirname(__file__), "data", "surfel_area_spacing111.npy")),
                                     dtype=np.float64)
        if _AREA.shape != (256,):
            raise RuntimeError("surfel area table is corrupt")
    return _AREA


def surface_distance_sums(mask_gt, mask_pred):
    """float64 device tensor [sum(d*area) gt->pred, sum(area) gt, sum(d*area) pred->gt, sum(area) pred]."""
    for t in (mask_gt, mask_pred):
        if not t.is_cuda or t.dtype != torch.uint8 or t.dim() != 3:
            raise RuntimeError("surface distance: masks must be 3-D uint8 ROCm device tensors (got %s %s on %s); there is no "
                               "CPU fallback" % (t.dtype, tuple(t.shape), t.device))
    if mask_gt.shape != mask_pred.shape:
        raise RuntimeError("surface distance: shapes differ %s vs %s" % (tuple(mask_gt.shape), tuple(mask_pred.shape)))
    L = _lib.lib()
    gt, pred = mask_gt.contiguous(), mask_pred.contiguous()
    d, h, w = (int(v) for v in gt.shape)
    out = torch.empty(4, dtype=torch.float64, device=gt.device)
    ws = _workspace(L.mri3d_surface_distance_workspace_bytes(d, h, w), gt.device)
    tab = _area_table()
    _lib.check(L.mri3d_surface_distance(_ptr(gt), _ptr(pred), d, h, w, tab.ctypes.data_as(ctypes.c_void_p), _ptr(out), _ptr(ws),
                                        ws.numel(), _stream()), "surface_distance")
    return out


def average_surface_distance(mask_gt, mask_pred):
    """(average distance gt -> pred, average distance pred -> gt) as numpy float64, like compute_average_surface_distance
    (metrics.py:180-207): NaN where a mask has no surface (0/0), inf where the other one has none."""
    s = surface_distance_sums(mask_gt, mask_pred).cpu().numpy()
    with np.errstate(invalid="ignore", divide="ignore"):
        return s[0] / s[1], s[2] / s[3]
