"""Patch pipeline on one MI355X: cutting training batches of 16 x 64^3 windows out of HBM-resident
subjects, and the grid inference of pretraining_3d_unet.ipynb cell 26 (36 windows of 64^3, overlap 4, U-Net c0=8, arg-max,
aggregate) on a 160x192x160 volume; the numpy restatement of the same window copies is timed beside it on one host core.
    python tests/perf/patch_bench.py"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mri_epilepsy_diagnosis_amd.segmentation import patches as P  # noqa: E402
from mri_epilepsy_diagnosis_amd.unet import UNet  # noqa: E402
from oracle import patches as O  # noqa: E402


def timed(fn, n, repeats=3):
    """best of `repeats` averages over n calls (ms per call), after a warm-up pass"""
    best = float("inf")
    for r in range(repeats + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        if r:
            best = min(best, (time.perf_counter() - t0) / n * 1e3)
    return best


shape, patch = (160, 192, 160), (64, 64, 64)
g = torch.Generator().manual_seed(0)
subjects = [{P.MRI: {P.DATA: torch.randn((1,) + shape, generator=g).cuda()},
             P.LABEL: {P.DATA: (torch.rand((1,) + shape, generator=g) < 0.1).float().cuda()}} for _ in range(16)]
q = P.Queue(subjects, max_length=240, samples_per_volume=8, patch_size=64, seed=0)
it = {"gen": q.batches(16)}


def next_batch():
    try:
        return next(it["gen"])
    except StopIteration:
        it["gen"] = q.batches(16)
        return next(it["gen"])


ms = timed(next_batch, 50)
mb = 16 * 64 ** 3 * 4 * 2 * 2 / 1e6  # image + label, read + write
print("queue batch (16 x 64^3 image+label windows): %.3f ms/batch = %.0f patches/s, %.1f GB/s of window copies"
      % (ms, 16 / ms * 1e3, mb / ms))

# the two extract launches alone (no host-side queue bookkeeping)
table = q._pop(16)
ms_k = timed(lambda: [P.extract_patches(v, table, patch) for v in q._resident.values()], 200)
print("  extract kernels only: %.3f ms/batch, %.1f GB/s" % (ms_k, mb / ms_k))
vol_np = subjects[0][P.MRI][P.DATA][0].cpu().numpy()
loc_np = np.concatenate([table[:, 1:], table[:, 1:] + 64], axis=1)
t0 = time.perf_counter()
for _ in range(5):
    O.extract(vol_np, loc_np), O.extract(vol_np, loc_np)
print("  numpy slicing of the same 2 x 16 windows on 1 core: %.2f ms/batch" % ((time.perf_counter() - t0) / 5 * 1e3))

torch.manual_seed(0)
model = UNet(in_channels=1, out_classes=2, dimensions=3, num_encoding_blocks=3, out_channels_first_layer=8,
             normalization="batch", upsampling_type="linear", padding=True, activation="PReLU").cuda().eval()
sample = subjects[0]
sampler = P.GridSampler(sample, 64, 4)


def grid_inference(batch):
    agg = P.GridAggregator(sample, 4)
    with torch.no_grad():
        for b in sampler.batches(batch):
            agg.add_batch_logits(model(b[P.MRI][P.DATA]), b[P.LOCATION])
    return agg.get_output_tensor(torch.uint8)


def copies_only(batch):
    agg = P.GridAggregator(sample, 4)
    for b in sampler.batches(batch):
        agg.add_batch(b[P.MRI][P.DATA].view(torch.uint8)[..., ::4].contiguous(), b[P.LOCATION])


for batch in (16, 36):
    print("grid inference 160x192x160, %d windows of 64^3 overlap 4, batch %d: %.2f ms/volume"
          % (len(sampler), batch, timed(lambda: grid_inference(batch), 10)))
with torch.no_grad():
    x = sample[P.MRI][P.DATA][None]
    print("whole-volume inference of the same model for comparison: %.2f ms/volume" % timed(lambda: model(x), 10))
wins = np.zeros((len(sampler),) + patch, np.uint8)
t0 = time.perf_counter()
for _ in range(3):
    O.aggregate(shape, wins, sampler.locations, (4, 4, 4))
print("numpy aggregate of 36 label windows on 1 core: %.2f ms/volume" % ((time.perf_counter() - t0) / 3 * 1e3))
loc = sampler.locations
lab = torch.zeros((len(sampler), 1) + patch, dtype=torch.uint8, device="cuda")
agg = P.GridAggregator(sample, 4)
print("device aggregate of 36 label windows: %.3f ms/volume" % timed(lambda: agg.add_batch(lab, loc), 100))
