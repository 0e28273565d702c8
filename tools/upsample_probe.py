"""Times trilinear x2 upsample forward/backward through the C ABI:  python tools/upsample_probe.py [--lib SO] C D H W [N] [f32|bf16]
(coarse size D H W; MRI3D_UP_GENERIC=1 selects the generic gather kernels for an A/B)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mri_epilepsy_diagnosis_amd import _lib, ops  # noqa: E402

if "--lib" in sys.argv:   # a tuning build, e.g. the previous commit as variant "prev"
    i = sys.argv.index("--lib")
    _lib.LIB_PATH = os.path.abspath(sys.argv[i + 1])
    del sys.argv[i:i + 2]

c, d, h, w = (int(a) for a in sys.argv[1:5])
n = int(sys.argv[5]) if len(sys.argv) > 5 else 2
dt = torch.bfloat16 if (len(sys.argv) > 6 and sys.argv[6] == "bf16") else torch.float32
x = torch.randn(n, c, d, h, w, device="cuda").to(dt).contiguous(memory_format=torch.channels_last_3d).requires_grad_(True)
y = ops.upsample3d(x, scale_factor=2, mode="trilinear", align_corners=False)
dy = torch.randn_like(y)


def timeit(fn, reps=10):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


es = x.element_size()
fwd = timeit(lambda: ops.upsample3d(x.detach(), scale_factor=2, mode="trilinear", align_corners=False))
bwd = timeit(lambda: torch.autograd.grad(y, x, dy, retain_graph=True))
byt = es * (x.numel() + y.numel())
print("upsample c%d %dx%dx%d n%d %s: fwd %.3f ms (%.0f GB/s)  bwd %.3f ms (%.0f GB/s)" % (
    c, d, h, w, n, str(dt)[6:], fwd, byt / fwd / 1e6, bwd, byt / bwd / 1e6))
