"""CPU restatement of the reference's losses (test oracle).

dice: segmentation/routine.py:239-253 (get_dice_score / get_dice_loss) applied as routine.py:272-274.
adversarial: classification/train_ENC_CLF.ipynb cell 14 (adv_loss / main_loss).
"""
import torch
import torch.nn.functional as F


def dice_score(output, target, dims=(2, 3, 4), epsilon=1e-9):
    p0, g0 = output, target
    p1, g1 = 1 - p0, 1 - g0
    tp = (p0 * g0).sum(dim=dims)
    fp = (p0 * g1).sum(dim=dims)
    fn = (p1 * g0).sum(dim=dims)
    return 2 * tp / (2 * tp + fp + fn + epsilon)


def softmax_dice_loss(logits, target):
    """F.softmax(dim=1) -> 1 - dice -> mean; a (N,1,...) target broadcasts against both classes."""
    return (1 - dice_score(F.softmax(logits, dim=1), target)).mean()


def iou_score(prediction, ground_truth):
    import numpy as np
    inter = np.logical_and(prediction > 0, ground_truth > 0).astype(np.float32).sum()
    union = np.logical_or(prediction > 0, ground_truth > 0).astype(np.float32).sum()
    return float(inter) / union


def adv_loss(domain, pred_logits, n_domains):
    onehot = torch.zeros((domain.shape[0], n_domains), dtype=torch.int32)
    onehot.scatter_(1, domain.view(-1, 1).cpu(), 1)
    rev = (1 - onehot).to(pred_logits.device)
    return -torch.mean(rev * F.log_softmax(pred_logits, dim=1))
