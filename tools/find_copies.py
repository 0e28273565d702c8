"""Which Python lines launch the torch copy / add kernels inside a U-Net step?  python tools/find_copies.py
(diagnostic: lists aten::copy_/add ops by device time with their call stacks)."""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mri_epilepsy_diagnosis_amd import ops, parallel  # noqa: E402
from mri_epilepsy_diagnosis_amd.unet import UNet  # noqa: E402

dev = torch.device("cuda")
torch.manual_seed(0)
net = UNet(in_channels=1, out_classes=2, dimensions=3, num_encoding_blocks=3, out_channels_first_layer=8,
           normalization="batch", upsampling_type="linear", padding=True, activation="PReLU").to(dev)
flat = parallel.FlatParams(net)
fopt = parallel.FlatAdam(flat, lr=1e-3, weight_decay=0.01, decoupled=True)
x = torch.randn(2, 1, 160, 192, 160, device=dev)
t = (torch.rand(2, 1, 160, 192, 160, device=dev) < 0.1).float()


def step():
    flat.zero_grad()
    ops.softmax_dice_loss(net(x), t).backward()
    fopt.step(flat.all_reduce())


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
evs = [e for e in prof.events() if e.name in ("aten::copy_", "aten::add", "aten::add_", "aten::clone", "aten::contiguous")]
evs.sort(key=lambda e: -(e.device_time_total if hasattr(e, "device_time_total") else e.cuda_time_total))
for e in evs[:12]:
    dt = e.device_time_total if hasattr(e, "device_time_total") else e.cuda_time_total
    print("%-16s %8.1f us  shapes=%s" % (e.name, dt, getattr(e, "input_shapes", None)))
    for fr in (e.stack or [])[:8]:
        print("      ", fr)
print("counts:", {n: sum(1 for e in prof.events() if e.name == n) for n in ("aten::copy_", "aten::add", "aten::add_", "aten::clone")})
