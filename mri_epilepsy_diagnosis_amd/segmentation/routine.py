"""Train / validate loop of the segmentation U-Net — host-side mirror of the reference's segmentation/routine.py
with the same public names, argument order and call sequence, driving the HIP operator set.

  reference symbol (segmentation/routine.py)            here
  Action :74, prepare_batch :185, get_iou_score :198     same semantics
  get_dice_score :239 / get_dice_loss :250               kept as tensor-formula helpers (API compatibility)
  forward :255, run_epoch :261, train :296               same call order: zero_grad -> forward -> softmax+dice ->
                                                         backward -> step -> .item(); initial VALIDATE epoch;
                                                         scheduler.step(mean val loss); save every `save_epoch`
  validate_dsc_asd :217                                  argmax -> uint8 mask on device (one fused kernel), Dice/IoU
  get_model_and_optimizer :338                           seeds, UNet(...), AdamW defaults, ReduceLROnPlateau

Differences, all deliberate: (1) F.softmax + get_dice_loss + .mean() (three full-resolution passes) is ONE fused HIP
kernel (`ops.softmax_dice_loss`); (2) the TorchIO/NIfTI loader builders (`get_loaders`, :97-183) are out of scope
(disk I/O) and raise; (3) surface-distance metrics (metrics.py, CPU scipy) are optional via the
`surface_metrics` hook; (4) no experiment tracker dependency (the `experiment` argument is honoured if given).
"""
import enum
import os
import time
import warnings

import numpy as np
import torch

from .. import ops, parallel
from . import surface
from ..unet import UNet

device = torch.device("cuda") if torch.cuda.is_available() else "cpu"
CHANNELS_DIMENSION = 1
SPATIAL_DIMENSIONS = 2, 3, 4

MRI = "MRI"
LABEL = "LABEL"
DATA = "data"  # torchio.DATA

LIST_FCD = [8, 10, 11, 12, 13, 16, 17, 18, 26, 47, 49, 50, 51, 52, 53, 54, 58, 85, 251, 252, 253, 254, 255]


class Action(enum.Enum):
    TRAIN = "Training"
    VALIDATE = "Validation"


def get_loaders(*args, **kwargs):
    raise NotImplementedError(
        "get_loaders builds TorchIO NIfTI datasets/queues (reference segmentation/routine.py:97-183); disk I/O is "
        "outside the MI355X hot path.  Feed run_epoch/train any iterable of {MRI: {DATA: x}, LABEL: {DATA: y}} batches.")


def prepare_batch(batch, device):
    """Move the volume to the device and binarise the label map as the reference does (routine.py:185-196):
    FreeSurfer ids in LIST_FCD -> 1 (first sample only, as in the reference), ids >= 1000 -> 1, everything else -> 0."""
    inputs = batch[MRI][DATA].to(device)
    targets = batch[LABEL][DATA]
    first = targets[0][0]
    first[torch.from_numpy(np.isin(first.cpu().numpy(), LIST_FCD))] = 1
    targets[targets >= 1000] = 1
    targets[targets != 1] = 0
    return inputs, targets.to(device)


def get_iou_score(prediction, ground_truth):
    intersection = np.logical_and(prediction > 0, ground_truth > 0).astype(np.float32).sum()
    union = np.logical_or(prediction > 0, ground_truth > 0).astype(np.float32).sum()
    return float(intersection) / union


def compute_dice_coefficient(mask_gt, mask_pred):
    """Volumetric Dice of two boolean masks; NaN when both are empty (segmentation/metrics.py:312-329)."""
    volume_sum = mask_gt.sum() + mask_pred.sum()
    if volume_sum == 0:
        return float("nan")
    return 2.0 * (mask_gt & mask_pred).sum() / volume_sum


def calculate_metrics(surface, prediction, surface_metrics=None):
    dsc = compute_dice_coefficient(surface.astype(bool), prediction.astype(bool))
    if surface_metrics is not None:
        asd_mean, asd_std = surface_metrics(surface, prediction)
    else:
        asd_mean = asd_std = float("nan")
    return dsc, asd_mean, asd_std, get_iou_score(prediction, surface)


def validate_dsc_asd(model, loader, surface_metrics=None):
    """Reference validate_dsc_asd (routine.py:217-237): per sample (dsc, asd gt->pred, asd pred->gt, iou).
    surface_metrics: None = on the device (default); False = skip (NaN); a callable(surface, prediction) -> (a, b) = host."""
    dsc, asd_mean, asd_std, iou = [], [], [], []
    model.eval()
    for batch in loader:
        inputs, targets = prepare_batch(batch, device)
        with torch.no_grad():
            logits = forward(model, inputs)
        labels = ops.argmax_mask(logits)  # (N, D, H, W) uint8 on device: no logits D2H
        if (surface_metrics is None or surface_metrics is False) and targets.is_cuda:
            # Dice / IoU from exact integer overlap counts and the two average surface distances from the on-device exact
            # distance transform: a few dozen bytes cross PCIe instead of two volumes (and no 5-7 s scipy EDT per volume)
            gt = targets[0][0].to(torch.uint8)      # .astype(np.uint8) of the reference
            d, i = ops.dice_iou_from_counts(ops.mask_overlap_counts(labels[0], gt))
            if surface_metrics is False:
                am = asd = float("nan")
            else:
                am, asd = surface.average_surface_distance(gt, labels[0])
        else:
            prediction = labels[0].cpu().numpy()
            d, am, asd, i = calculate_metrics(targets.cpu().numpy().astype(np.uint8)[0][0], prediction, surface_metrics)
        dsc.append(d), asd_mean.append(am), asd_std.append(asd), iou.append(i)
    return dsc, asd_mean, asd_std, iou


def get_dice_score(output, target, SPATIAL_DIMENSIONS=(2, 3, 4), epsilon=1e-9):
    """Tensor-formula helper kept for API compatibility; the training loop uses the fused HIP loss instead."""
    p0, g0 = output, target
    tp = (p0 * g0).sum(dim=SPATIAL_DIMENSIONS)
    fp = (p0 * (1 - g0)).sum(dim=SPATIAL_DIMENSIONS)
    fn = ((1 - p0) * g0).sum(dim=SPATIAL_DIMENSIONS)
    return 2 * tp / (2 * tp + fp + fn + epsilon)


def get_dice_loss(output, target):
    return 1 - get_dice_score(output, target)


def forward(model, inputs):
    with warnings.catch_warnings():
        warnings.filterwarnings("ignore", category=UserWarning)
        logits = model(inputs)
    return logits


def run_epoch(epoch_idx, action, loader, model, optimizer, scheduler=False, experiment=False, loss_fn=None):
    """One pass over `loader` (segmentation/routine.py:261-294: zero_grad -> forward -> softmax-Dice -> backward -> step ->
    .item()).  `loss_fn(logits, targets)` defaults to the fused softmax+Dice HIP kernel.  On the device the sequence
    zero_grad -> forward -> loss (-> backward) is captured once per batch shape into a hipGraph and replayed
    (parallel.StepCache: same kernels, same results, one launch instead of ~300); CPU modules / tensors (the oracle-driven host
    tests) and anything that cannot be captured run eagerly."""
    is_training = action == Action.TRAIN
    loss_fn = ops.softmax_dice_loss if loss_fn is None else loss_fn
    epoch_losses = []
    model.train(is_training)
    cache = parallel.StepCache.of(model)
    for batch in loader:
        inputs, targets = prepare_batch(batch, device)
        optimizer.zero_grad()
        with torch.set_grad_enabled(is_training), warnings.catch_warnings():
            warnings.filterwarnings("ignore", category=UserWarning)
            logits, batch_loss = cache.run(inputs, targets, loss_fn, backward=is_training)
            if is_training:
                optimizer.step()
            epoch_losses.append(batch_loss.item())
            if experiment:
                name = "train_dice_loss" if is_training else "validate_dice_loss"
                experiment.log_metric(name, epoch_losses[-1])
        del inputs, targets, logits, batch_loss
    return np.array(epoch_losses)


def train(num_epochs, training_loader, validation_loader, model, optimizer, scheduler, weights_stem, save_epoch=1,
          experiment=False, verbose=True, loss_fn=None, weights_dir="weights"):
    start_time = time.time()
    epoch_train_loss, epoch_val_loss = [], []
    run_epoch(0, Action.VALIDATE, validation_loader, model, optimizer, scheduler, experiment, loss_fn)
    for epoch_idx in range(1, num_epochs + 1):
        epoch_train_losses = run_epoch(epoch_idx, Action.TRAIN, training_loader, model, optimizer, scheduler,
                                       experiment, loss_fn)
        epoch_val_losses = run_epoch(epoch_idx, Action.VALIDATE, validation_loader, model, optimizer, scheduler,
                                     experiment, loss_fn)
        if verbose:
            print("Epoch {} of {} took {:.3f}s".format(epoch_idx, num_epochs, time.time() - start_time))
            print("  training loss (in-iteration): \t{:.6f}".format(epoch_train_losses[-1]))
            print("  validation loss: \t\t\t{:.6f}".format(epoch_val_losses[-1]))
        epoch_train_loss.append(np.mean(epoch_train_losses))
        epoch_val_loss.append(np.mean(epoch_val_losses))
        if scheduler:
            scheduler.step(np.mean(epoch_val_losses))
        if experiment:
            experiment.log_epoch_end(epoch_idx)
        if epoch_idx % save_epoch == 0:
            os.makedirs(weights_dir, exist_ok=True)
            torch.save(model.state_dict(), os.path.join(weights_dir, f"{weights_stem}_epoch_{epoch_idx}.pth"))
    return epoch_train_loss, epoch_val_loss


def get_model_and_optimizer(device, num_encoding_blocks=3, out_channels_first_layer=16, patience=3):
    torch.manual_seed(0)
    np.random.seed(0)
    model = UNet(in_channels=1, out_classes=2, dimensions=3, num_encoding_blocks=num_encoding_blocks,
                 out_channels_first_layer=out_channels_first_layer, normalization="batch", upsampling_type="linear",
                 padding=True, activation="PReLU").to(device)
    optimizer = torch.optim.AdamW(model.parameters())
    scheduler = torch.optim.lr_scheduler.ReduceLROnPlateau(optimizer, mode="min", factor=0.1, patience=patience,
                                                           threshold=0.01)
    return model, optimizer, scheduler


def synthetic_loader(n_batches, batch_size, shape, device="cpu", seed=1234, foreground=0.1):
    """Synthetic z-normalised-T1-like volumes + Bernoulli(foreground) label maps in the reference's
    batch-dict format.  Labels are 0/1 floats already, so prepare_batch's binarisation is the identity on them."""
    g = torch.Generator().manual_seed(seed)
    batches = []
    for _ in range(n_batches):
        x = torch.randn(batch_size, 1, *shape, generator=g)
        y = (torch.rand(batch_size, 1, *shape, generator=g) < foreground).float()
        batches.append({MRI: {DATA: x.to(device)}, LABEL: {DATA: y.to(device)}})
    return batches
