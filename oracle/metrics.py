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
