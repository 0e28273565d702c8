"""CPU restatement of the reference's mask-overlap validation metrics (TEST ORACLE — only tests/, smoke() and bench.py's
cpu_baseline may import this package).

dice_coefficient: segmentation/metrics.py:312-329 (compute_dice_coefficient; NaN when both masks are empty).
iou_score:        segmentation/routine.py:198-203 (get_iou_score).
Pinned against the reference functions themselves by oracle/gen_golden.py (tests/golden/mask_metrics.npz).
"""
import numpy as np


def dice_coefficient(mask_gt, mask_pred):
    volume_sum = mask_gt.sum() + mask_pred.sum()
    if volume_sum == 0:
        return float("nan")
    volume_intersect = (mask_gt & mask_pred).sum()
    return 2 * volume_intersect / volume_sum


def iou_score(prediction, ground_truth):
    intersection = np.logical_and(prediction > 0, ground_truth > 0).astype(np.float32).sum()
    union = np.logical_or(prediction > 0, ground_truth > 0).astype(np.float32).sum()
    return float(intersection) / union


def seeded_masks(seed, shape, p_gt, p_pred, corr):
    """Two correlated uint8 {0,1} masks from numpy's PCG64 stream (deterministic for a fixed numpy build)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    base = rng.random(shape)
    noise = rng.random(shape)
    gt = (base < p_gt).astype(np.uint8)
    pred = (np.where(noise < corr, base, rng.random(shape)) < p_pred).astype(np.uint8)
    return gt, pred


def surface_elements(mask):
    """Neighbour codes of a binary mask on the (D+1,H+1,W+1) corner grid (metrics.py:100-127, without the bounding-box crop,
    which does not change any distance) and the surface-element flags (code not in {0, 255})."""
    from scipy import ndimage
    m = np.zeros(tuple(n + 1 for n in mask.shape), np.uint8)
    m[:-1, :-1, :-1] = mask != 0
    kernel = np.array([[[128, 64], [32, 16]], [[8, 4], [2, 1]]])
    code = ndimage.correlate(m, kernel, mode="constant", cval=0)
    return code, (code != 0) & (code != 255)


def average_surface_distance(mask_gt, mask_pred, area_table):
    """compute_average_surface_distance(compute_surface_distances(gt, pred, (1,1,1))) restated (metrics.py:25-207):
    area-weighted mean distance from each mask's surface elements to the other mask's surface (scipy EDT, unit spacing)."""
    from scipy import ndimage
    code_g, bord_g = surface_elements(mask_gt)
    code_p, bord_p = surface_elements(mask_pred)
    if not (bord_g.any() or bord_p.any()):
        return float("nan"), float("nan")
    dist_g = ndimage.distance_transform_edt(~bord_g) if bord_g.any() else np.inf * np.ones(bord_g.shape)
    dist_p = ndimage.distance_transform_edt(~bord_p) if bord_p.any() else np.inf * np.ones(bord_p.shape)
    d_gp, a_g = dist_p[bord_g], area_table[code_g][bord_g]
    d_pg, a_p = dist_g[bord_p], area_table[code_p][bord_p]
    with np.errstate(invalid="ignore", divide="ignore"):
        return np.sum(d_gp * a_g) / np.sum(a_g), np.sum(d_pg * a_p) / np.sum(a_p)


def seeded_blobs(seed, shape, thr_gt=0.0, thr_pred=0.15):
    """Two overlapping smooth blobs (uint8) from low-pass filtered PCG64 noise: realistic connected surfaces."""
    from scipy import ndimage
    rng = np.random.Generator(np.random.PCG64(seed))
    f = ndimage.gaussian_filter(rng.normal(size=shape), 2.5)
    g = f + 0.3 * ndimage.gaussian_filter(rng.normal(size=shape), 1.5)
    f /= f.std()
    g /= g.std()
    return (f > thr_gt).astype(np.uint8), (g > thr_pred).astype(np.uint8)


def surface_distances(mask_gt, mask_pred, area_table):
    """compute_surface_distances (metrics.py:25-178, unit spacing) restated with scipy: the sorted distance / area lists."""
    from scipy import ndimage
    code_g, bord_g = surface_elements(mask_gt)
    code_p, bord_p = surface_elements(mask_pred)
    if not (np.asarray(mask_gt).any() or np.asarray(mask_pred).any()):
        e = np.array([])
        return {"distances_gt_to_pred": e, "distances_pred_to_gt": e, "surfel_areas_gt": e, "surfel_areas_pred": e}
    dist_g = ndimage.distance_transform_edt(~bord_g) if bord_g.any() else np.inf * np.ones(bord_g.shape)
    dist_p = ndimage.distance_transform_edt(~bord_p) if bord_p.any() else np.inf * np.ones(bord_p.shape)
    out = {}
    for name, key, d, a in (("gt_to_pred", "gt", dist_p[bord_g], area_table[code_g][bord_g]),
                            ("pred_to_gt", "pred", dist_g[bord_p], area_table[code_p][bord_p])):
        if d.shape != (0,):
            srt = np.array(sorted(zip(d, a)))
            d, a = srt[:, 0], srt[:, 1]
        out["distances_" + name] = d
        out["surfel_areas_" + key] = a
    return out


def robust_hausdorff(sd, percent):
    """metrics.py:208-247."""
    res = []
    for dist, area in ((sd["distances_gt_to_pred"], sd["surfel_areas_gt"]), (sd["distances_pred_to_gt"], sd["surfel_areas_pred"])):
        if len(dist) > 0:
            cum = np.cumsum(area) / np.sum(area)
            res.append(dist[min(np.searchsorted(cum, percent / 100.0), len(dist) - 1)])
        else:
            res.append(np.inf)
    return max(res)


def surface_dice_at_tolerance(sd, tol):
    """metrics.py:280-309."""
    og = np.sum(sd["surfel_areas_gt"][sd["distances_gt_to_pred"] <= tol])
    op = np.sum(sd["surfel_areas_pred"][sd["distances_pred_to_gt"] <= tol])
    return (og + op) / (np.sum(sd["surfel_areas_gt"]) + np.sum(sd["surfel_areas_pred"]))
