"""MaxPool3d(2) forward / backward(+skip) timing on the U-Net's pooled tensors:  python tools/pool_bench.py [--lib SO]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mri_epilepsy_diagnosis_amd import _lib, ops  # noqa: E402

if "--lib" in sys.argv:
    i = sys.argv.index("--lib")
    _lib.LIB_PATH = os.path.abspath(sys.argv[i + 1])
    del sys.argv[i:i + 2]


def t(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


for dt in (torch.float32, torch.bfloat16):
    for c, sp in ((16, (160, 192, 160)), (32, (80, 96, 80)), (8, (160, 192, 160))):
        x = torch.randn(2, c, *sp, device="cuda").to(dt).contiguous(memory_format=torch.channels_last_3d).requires_grad_(True)
        y, skip = ops.max_pool3d_skip(x, 2)
        gy, gs = torch.randn_like(y), torch.randn_like(x)
        with torch.no_grad():
            f = t(lambda: ops.max_pool3d(x, 2))
        b = t(lambda: torch.autograd.backward([y, skip], [gy, gs], retain_graph=True))
        print("%s c%d %s: fwd %.3f ms  bwd+skip %.3f ms" % (str(dt)[6:], c, "x".join(map(str, sp)), f, b), flush=True)
