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


_SD_INF = 0x3f000000


def surface_distances(mask_gt, mask_pred):
    """The dict compute_surface_distances returns (metrics.py:25-178), built from the device's exact squared distances:
    per direction the distances sorted ascending (ties by area) with the matching surface-element areas, float64."""
    for t in (mask_gt, mask_pred):
        if not t.is_cuda or t.dtype != torch.uint8 or t.dim() != 3:
            raise RuntimeError("surface distances: masks must be 3-D uint8 ROCm device tensors; there is no CPU fallback")
    L = _lib.lib()
    gt, pred = mask_gt.contiguous(), mask_pred.contiguous()
    d, h, w = (int(v) for v in gt.shape)
    cap = (d + 1) * (h + 1) * (w + 1)
    dev = gt.device
    d2 = [torch.empty(cap, dtype=torch.int32, device=dev) for _ in range(2)]
    code = [torch.empty(cap, dtype=torch.uint8, device=dev) for _ in range(2)]
    counts = torch.empty(2, dtype=torch.int64, device=dev)
    ws = _workspace(L.mri3d_surface_distance_workspace_bytes(d, h, w), dev)
    _lib.check(L.mri3d_surface_elements(_ptr(gt), _ptr(pred), d, h, w, _ptr(d2[0]), _ptr(code[0]), _ptr(d2[1]), _ptr(code[1]),
                                        cap, _ptr(counts), _ptr(ws), ws.numel(), _stream()), "surface_elements")
    n = counts.cpu().tolist()
    tab = _area_table()
    out = {}
    for k, name in enumerate(("gt_to_pred", "pred_to_gt")):
        sq = d2[k][:n[k]].cpu().numpy()
        area = tab[code[k][:n[k]].cpu().numpy()]
        dist = np.where(sq >= _SD_INF, np.inf, np.sqrt(sq.astype(np.float64)))
        order = np.lexsort((area, dist))           # sorted(zip(distances, areas)) of the reference
        out["distances_" + name] = dist[order]
        out["surfel_areas_" + ("gt" if k == 0 else "pred")] = area[order]
    return out


def compute_average_surface_distance(surface_distances):
    """metrics.py:180-207."""
    sd = surface_distances
    with np.errstate(invalid="ignore", divide="ignore"):
        return (np.sum(sd["distances_gt_to_pred"] * sd["surfel_areas_gt"]) / np.sum(sd["surfel_areas_gt"]),
                np.sum(sd["distances_pred_to_gt"] * sd["surfel_areas_pred"]) / np.sum(sd["surfel_areas_pred"]))


def compute_robust_hausdorff(surface_distances, percent):
    """metrics.py:208-247: area-weighted `percent`-th percentile of the surface distances, the larger of both directions."""
    sd = surface_distances
    res = []
    for dist, area in ((sd["distances_gt_to_pred"], sd["surfel_areas_gt"]), (sd["distances_pred_to_gt"], sd["surfel_areas_pred"])):
        if len(dist) > 0:
            cum = np.cumsum(area) / np.sum(area)
            idx = np.searchsorted(cum, percent / 100.0)
            res.append(dist[min(idx, len(dist) - 1)])
        else:
            res.append(np.inf)
    return max(res)


def compute_surface_overlap_at_tolerance(surface_distances, tolerance_mm):
    """metrics.py:250-277."""
    sd = surface_distances
    with np.errstate(invalid="ignore", divide="ignore"):
        return (np.sum(sd["surfel_areas_gt"][sd["distances_gt_to_pred"] <= tolerance_mm]) / np.sum(sd["surfel_areas_gt"]),
                np.sum(sd["surfel_areas_pred"][sd["distances_pred_to_gt"] <= tolerance_mm]) / np.sum(sd["surfel_areas_pred"]))


def compute_surface_dice_at_tolerance(surface_distances, tolerance_mm):
    """metrics.py:280-309."""
    sd = surface_distances
    overlap_gt = np.sum(sd["surfel_areas_gt"][sd["distances_gt_to_pred"] <= tolerance_mm])
    overlap_pred = np.sum(sd["surfel_areas_pred"][sd["distances_pred_to_gt"] <= tolerance_mm])
    with np.errstate(invalid="ignore", divide="ignore"):
        return (overlap_gt + overlap_pred) / (np.sum(sd["surfel_areas_gt"]) + np.sum(sd["surfel_areas_pred"]))
