"""Which autograd nodes still launch torch elementwise kernels (aten::add / copy_) in a training step?  One eager U-Net step
under torch.profiler (CPU side only), grouped by operator, input shapes and the Python call site."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mri_epilepsy_diagnosis_amd import ops  # noqa: E402
from mri_epilepsy_diagnosis_amd.unet import UNet  # noqa: E402

bf16 = len(sys.argv) > 1 and sys.argv[1] == "bf16"
dev = torch.device("cuda")
torch.manual_seed(0)
net = UNet(in_channels=1, out_classes=2, dimensions=3, num_encoding_blocks=3, out_channels_first_layer=8,
           normalization="batch", upsampling_type="linear", padding=True, activation="PReLU").to(dev)
S = (160, 192, 160) if "full" in sys.argv else (64, 64, 64)
x = torch.randn(2, 1, *S, device=dev)
t = (torch.rand(2, 1, *S, device=dev) < 0.1).float()


from mri_epilepsy_diagnosis_amd import parallel  # noqa: E402

flat = parallel.FlatParams(net) if "flat" in sys.argv else None   # bench.py's setting: gradients land in one flat buffer
fopt = parallel.FlatAdam(flat, lr=1e-3, weight_decay=1e-2, decoupled=True) if flat is not None else None


def step():
    if flat is not None:
        flat.zero_grad()
    else:
        net.zero_grad(set_to_none=True)
    with ops.autocast(enabled=bf16):
        loss = ops.softmax_dice_loss(net(x), t)
    loss.backward()
    if flat is not None:
        fopt.step(flat.all_reduce())


step()
torch.cuda.synchronize()
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU], record_shapes=True, with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
for ev in prof.events():
    if ev.name.startswith("aten::") and ev.input_shapes:
        big = [s for s in ev.input_shapes if s and len(s) >= 4]
        if not big and "small" not in sys.argv:
            continue
        if "small" in sys.argv and ev.name not in ("aten::copy_", "aten::add_", "aten::add", "aten::zero_", "aten::fill_", "aten::clone",
                                                    "aten::mul_", "aten::to", "aten::_to_copy", "aten::sub", "aten::mul", "aten::div"):
            continue
        stack = [s for s in (ev.stack or []) if "mri_epilepsy" in s or "autograd" in s][:3]
        print(ev.name, ev.input_shapes, "|", " <- ".join(stack))
