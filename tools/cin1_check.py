"""First-layer Conv3d(1, Co, 3, padding=1) kernels against torch on the CPU (fp64) for a few ragged shapes, then timing
through tools/conv_bench.py's entry.   python tools/cin1_check.py"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mri_epilepsy_diagnosis_amd import ops  # noqa: E402

torch.manual_seed(0)
worst = 0.0
for co, shape, bias in ((8, (2, 1, 9, 13, 37), True), (16, (1, 1, 8, 16, 32), False), (8, (1, 1, 4, 8, 32), False),
                        (16, (3, 1, 5, 7, 19), True), (8, (1, 1, 1, 1, 1), True), (8, (1, 1, 17, 9, 70), True),
                        (1, (2, 1, 9, 13, 37), True), (1, (1, 1, 4, 8, 32), False), (1, (1, 1, 1, 1, 1), True),
                        (1, (2, 1, 17, 9, 70), True)):
    x = torch.randn(shape)
    w = torch.randn(co, 1, 3, 3, 3) * 0.2
    b = torch.randn(co) if bias else None
    xd = x.cuda().requires_grad_(True)
    wd = w.cuda().requires_grad_(True)
    bd = b.cuda().requires_grad_(True) if bias else None
    y = ops.conv3d(xd, wd, bd, padding=1)
    dy = torch.randn(y.shape)
    y.backward(dy.cuda())
    x64, w64 = x.double().requires_grad_(True), w.double().requires_grad_(True)
    b64 = b.double().requires_grad_(True) if bias else None
    r = F.conv3d(x64, w64, b64, padding=1)
    r.backward(dy.double())
    errs = [(y.detach().cpu().double() - r).abs().max().item(), (wd.grad.cpu().double() - w64.grad).abs().max().item(),
            (xd.grad.cpu().double() - x64.grad).abs().max().item()]
    if bias:
        errs.append((bd.grad.cpu().double() - b64.grad).abs().max().item())
    scale = max(1.0, w64.grad.abs().max().item())
    print("co=%d shape=%s bias=%s  max|err| y %.2e  dw %.2e (|dw| up to %.1f)  dx %.2e %s" % (
        co, shape, bias, errs[0], errs[1], scale, errs[2], ("db %.2e" % errs[3]) if bias else ""))
    worst = max(worst, errs[0], errs[1] / scale, errs[2])
print("worst relative-ish error %.2e" % worst)
assert worst < 1e-4
