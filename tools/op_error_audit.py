"""Print the max-norm relative error of each HIP op (fwd and grads) against an fp64 CPU evaluation, next to the
error of torch's fp32 CPU path — a diagnostic for hunting precision outliers (run on the GPU box)."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mri_epilepsy_diagnosis_amd import ops  # noqa: E402

DEV = "cuda"


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-300)).item()


def cl(t):
    return t.to(DEV).contiguous(memory_format=torch.channels_last_3d) if t.dim() == 5 else t.to(DEV)


def run(name, fn_ref, fn_hip, inputs):
    """inputs: list of fp32 CPU tensors (leaf).  fn_* take the list and return a tensor."""
    res = {}
    for tag, cast in (("f64", lambda t: t.double()), ("f32", lambda t: t.clone())):
        xs = [cast(t).requires_grad_(True) for t in inputs]
        y = fn_ref(xs)
        g = torch.randn(y.shape, generator=torch.Generator().manual_seed(7), dtype=torch.float32).to(y.dtype)
        y.backward(g)
        res[tag] = (y, [x.grad for x in xs])
    xs = [cl(t).requires_grad_(True) for t in inputs]
    y = fn_hip(xs)
    g = torch.randn(y.shape, generator=torch.Generator().manual_seed(7), dtype=torch.float32)
    y.backward(cl(g))
    y64, g64 = res["f64"]
    y32, g32 = res["f32"]
    line = "%-34s y: hip %.1e cpu %.1e |" % (name, rel(y, y64), rel(y32, y64))
    for i, (gh, gc, gt) in enumerate(zip([x.grad for x in xs], g32, g64)):
        line += " g%d: hip %.1e cpu %.1e |" % (i, rel(gh, gt), rel(gc, gt))
    print(line, flush=True)


torch.manual_seed(0)
R = lambda *s: torch.randn(*s)

for (ci, co, sp, k, s, p, d) in [(16, 16, (16, 16, 16), 3, 1, 1, 1), (48, 16, (8, 16, 16), 3, 1, 1, 1), (96, 32, (8, 8, 16), 3, 1, 1, 1),
                                 (8, 16, (12, 16, 16), 3, 1, 1, 1), (1, 8, (16, 16, 16), 3, 1, 1, 1), (8, 16, (16, 16, 16), 3, 2, 1, 1),
                                 (16, 2, (16, 16, 16), 1, 1, 0, 1), (8, 8, (32, 8, 8), (6, 1, 1), (2, 1, 1), (2, 0, 0), 1)]:
    run("conv %d->%d k%s s%s" % (ci, co, k, s), lambda xs: F.conv3d(xs[0], xs[1], xs[2], s, p, d),
        lambda xs: ops.conv3d(xs[0], xs[1], xs[2], s, p, d), [R(2, ci, *sp), R(co, ci, *((k,) * 3 if isinstance(k, int) else k)) * 0.1, R(co)])

for mode in ("batch", "instance"):
    for sp in ((16, 16, 16), (2, 2, 2), (4, 4, 4)):
        x = R(2, 16, *sp) * 2 + 1.5
        if mode == "batch":
            ref = lambda xs: F.prelu(F.batch_norm(xs[0], None, None, xs[1], xs[2], True, 0.1, 1e-5), xs[3])
        else:
            ref = lambda xs: F.prelu(F.instance_norm(xs[0], None, None, xs[1], xs[2], True, 0.1, 1e-5), xs[3])
        run("norm %s %s + prelu" % (mode, sp), ref,
            lambda xs: ops.norm_act(xs[0], xs[1], xs[2], xs[3], None, None, mode, 0.1, 1e-5, "prelu", 0.0),
            [x, R(16) * 0.3 + 1, R(16) * 0.2, torch.tensor([0.25])])
    x = R(2, 16, 8, 8, 8)
    run("norm %s none-affine lrelu" % mode,
        (lambda xs: F.leaky_relu(F.batch_norm(xs[0], None, None, None, None, True, 0.1, 1e-5), 0.01)) if mode == "batch" else
        (lambda xs: F.leaky_relu(F.instance_norm(xs[0], None, None, None, None, True, 0.1, 1e-5), 0.01)),
        lambda xs: ops.norm_act(xs[0], None, None, None, None, None, mode, 0.1, 1e-5, "leaky_relu", 0.01), [x])

run("maxpool2", lambda xs: F.max_pool3d(xs[0], 2), lambda xs: ops.max_pool3d(xs[0], 2), [R(2, 16, 8, 8, 8)])
run("upsample trilinear x2", lambda xs: F.interpolate(xs[0], scale_factor=2, mode="trilinear", align_corners=False),
    lambda xs: ops.upsample3d(xs[0], scale_factor=2, mode="trilinear", align_corners=False), [R(2, 16, 6, 6, 6)])
run("upsample nearest x2", lambda xs: F.interpolate(xs[0], scale_factor=2, mode="nearest"),
    lambda xs: ops.upsample3d(xs[0], scale_factor=2, mode="nearest"), [R(2, 16, 6, 6, 6)])
tgt = (torch.rand(2, 1, 12, 12, 12) < 0.2).float()


def dice_ref(xs):
    p = F.softmax(xs[0], dim=1)
    g0 = tgt.to(p.dtype)
    tp = (p * g0).sum(dim=(2, 3, 4)); fp = (p * (1 - g0)).sum(dim=(2, 3, 4)); fn = ((1 - p) * g0).sum(dim=(2, 3, 4))
    return (1 - 2 * tp / (2 * tp + fp + fn + 1e-9)).mean().reshape(1)


run("softmax dice", dice_ref, lambda xs: ops.softmax_dice_loss(xs[0], cl(tgt)).reshape(1), [R(2, 2, 12, 12, 12)])
run("add", lambda xs: xs[0] + xs[1], lambda xs: ops.add(xs[0], xs[1]), [R(2, 8, 4, 4, 4), R(2, 8, 4, 4, 4)])
