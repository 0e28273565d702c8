"""What does `parallel.SyncBatchNorm` (the parity mode of SURVEY §8e) cost per step?  One rank, the bench's U-Net step
(2 x 160x192x160 fp32, fwd + soft-Dice + bwd + AdamW), eager, with and without the context: the collectives are no-ops at
world size 1, so the difference is the mode's own kernel path (statistics as shifted moments in float64 on the host stream,
frozen-statistics backward + the per-channel correction pass) — an upper bound for what N ranks add on top of 32 tiny
all-reduces per step."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mri_epilepsy_diagnosis_amd import ops, parallel  # noqa: E402
from mri_epilepsy_diagnosis_amd.unet import UNet  # noqa: E402

dev = torch.device("cuda")
torch.manual_seed(0)
net = UNet(in_channels=1, out_classes=2, dimensions=3, num_encoding_blocks=3, out_channels_first_layer=8, normalization="batch",
           upsampling_type="linear", padding=True, activation="PReLU").to(dev).train()
flat = parallel.FlatParams(net)
opt = parallel.FlatAdam(flat)
g = torch.Generator(device=dev).manual_seed(1)
x = torch.randn(2, 1, 160, 192, 160, device=dev, generator=g)
t = (torch.rand(2, 1, 160, 192, 160, device=dev, generator=g) < 0.1).float()


def step(sync):
    flat.zero_grad()
    if sync:
        with parallel.SyncBatchNorm():
            loss = ops.softmax_dice_loss(net(x), t)
            loss.backward()
    else:
        loss = ops.softmax_dice_loss(net(x), t)
        loss.backward()
    opt.step(flat.all_reduce())


for sync in (False, True, False, True):
    for _ in range(3):
        step(sync)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        step(sync)
    torch.cuda.synchronize()
    print("SyncBatchNorm %-5s  %.2f ms/step" % (sync, (time.perf_counter() - t0) * 100), flush=True)
