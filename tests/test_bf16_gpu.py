"""bf16-storage path (BASELINE configs[3], SURVEY §8 a1 "bf16 storage + fp32 accumulate"): every volumetric op run on
bfloat16 activations through the C-ABI must equal the fp32 oracle arithmetic applied to the SAME bf16-rounded inputs,
up to the final round-to-nearest-even of the stored result (1 bf16 ulp = 2^-8 relative; tolerance 2 ulp + a small
absolute term for values that cancel).  Parameters / statistics / parameter gradients stay fp32."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

BF = torch.bfloat16


def _ops():
    from mri_epilepsy_diagnosis_amd import ops
    return ops


def _close(got, ref, what, ulp=2.0, abs_frac=2e-3):
    got = got.detach().float().cpu()
    ref = ref.detach().float().cpu()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    scale = ref.abs().max().item() + 1e-30
    err = (got - ref).abs()
    tol = ulp * 2.0 ** -8 * ref.abs() + abs_frac * scale
    bad = (err > tol)
    assert not bad.any(), "%s: %d/%d outside tolerance, max err %.3e (scale %.3e)" % (
        what, int(bad.sum()), bad.numel(), err.max().item(), scale)


def _rb(t):
    """round to bf16 and back (what the device stores)."""
    return t.to(BF).float()


def _dev(t, dtype=BF):
    return t.to("cuda").to(dtype).contiguous(memory_format=torch.channels_last_3d) if t.dim() == 5 else t.to("cuda")


CONV_CASES = [
    # (n, ci, co, size, k, stride, pad, dil)
    (2, 8, 16, (12, 20, 18), 3, 1, 1, 1),       # MFMA-shaped 3x3x3
    (1, 16, 16, (9, 17, 33), 3, 1, 1, 1),
    (1, 48, 16, (8, 16, 16), 3, 1, 1, 1),
    (2, 32, 64, (6, 9, 17), 3, 1, 1, 1),
    (1, 96, 32, (5, 8, 16), 3, 1, 1, 1),
    (2, 1, 8, (10, 12, 14), 3, 1, 1, 1),        # first layer
    (2, 16, 2, (10, 12, 14), 1, 1, 0, 1),       # classifier
    (1, 8, 16, (11, 12, 13), 3, 2, 1, 1),       # strided (Modified3DUNet)
    (1, 4, 6, (9, 10, 11), (3, 1, 1), (2, 1, 1), (1, 0, 0), 1),
]


@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: "n%d_%d-%d_%s_k%s_s%s" % (c[0], c[1], c[2], "x".join(map(str, c[3])), c[4], c[5]))
def test_conv3d_bf16(case):
    ops = _ops()
    n, ci, co, size, k, stride, pad, dil = case
    torch.manual_seed(11)
    x = _rb(torch.randn(n, ci, *size))
    conv = torch.nn.Conv3d(ci, co, k, stride=stride, padding=pad, dilation=dil)
    w, b = conv.weight.detach(), conv.bias.detach()
    wq = _rb(w)                                  # the kernels round weights to bf16 for the MFMA operands
    xr = x.clone().requires_grad_(True)
    wr = wq.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True)
    yr = F.conv3d(xr, wr, br, stride=stride, padding=pad, dilation=dil)
    dy = _rb(torch.randn_like(yr))
    yr.backward(dy)

    xg = _dev(x).requires_grad_(True)
    wg = w.cuda().requires_grad_(True)
    bg = b.cuda().requires_grad_(True)
    yg = ops.conv3d(xg, wg, bg, stride=stride, padding=pad, dilation=dil)
    assert yg.dtype == BF
    yg.backward(_dev(dy))
    assert xg.grad.dtype == BF and wg.grad.dtype == torch.float32 and bg.grad.dtype == torch.float32
    # generic (non-MFMA) kernels keep fp32 weights: compare against whichever weight rounding is closer
    y_alt = F.conv3d(x, w, b, stride=stride, padding=pad, dilation=dil)
    e1 = (yg.float().cpu() - yr.detach()).abs().max().item()
    e2 = (yg.float().cpu() - y_alt).abs().max().item()
    if e2 < e1:   # fp32-weight kernel
        xr2 = x.clone().requires_grad_(True)
        wr2 = w.clone().requires_grad_(True)
        br2 = b.clone().requires_grad_(True)
        yr2 = F.conv3d(xr2, wr2, br2, stride=stride, padding=pad, dilation=dil)
        yr2.backward(dy)
        yr, xr, wr, br = yr2, xr2, wr2, br2
    _close(yg, yr, "conv y")
    _close(xg.grad, xr.grad, "conv dx")
    _close(wg.grad, wr.grad, "conv dw", ulp=0.0, abs_frac=2e-3)
    _close(bg.grad, br.grad, "conv db", ulp=0.0, abs_frac=2e-3)


@pytest.mark.parametrize("mode,act", [("batch", "prelu"), ("instance", "leaky_relu"), ("batch", "relu"), ("none", "prelu")])
@pytest.mark.parametrize("c", [16, 6])
def test_norm_act_bf16(mode, act, c):
    ops = _ops()
    torch.manual_seed(5)
    x = _rb(torch.randn(2, c, 7, 9, 12) * 2.0 + 0.5)
    gamma = torch.rand(c) + 0.5
    beta = torch.randn(c) * 0.2
    alpha = torch.tensor([0.25])
    xr = x.clone().requires_grad_(True)
    gr, br, ar = (t.clone().requires_grad_(True) for t in (gamma, beta, alpha))
    if mode == "batch":
        h = F.batch_norm(xr, None, None, gr, br, True, 0.1, 1e-5)
    elif mode == "instance":
        h = F.instance_norm(xr, weight=None, bias=None, eps=1e-5)
    else:
        h = xr
    yr = {"prelu": lambda t: F.prelu(t, ar), "leaky_relu": lambda t: F.leaky_relu(t, 0.01), "relu": F.relu}[act](h)
    dy = _rb(torch.randn_like(yr))
    yr.backward(dy)

    xg = _dev(x).requires_grad_(True)
    gg, bg, ag = (t.cuda().requires_grad_(True) for t in (gamma, beta, alpha))
    affine = mode == "batch"
    yg = ops.norm_act(xg, gg if affine else None, bg if affine else None, ag if act == "prelu" else None, None, None, mode,
                      0.1, 1e-5, act, 0.01)
    assert yg.dtype == BF
    yg.backward(_dev(dy))
    _close(yg, yr, "norm y")
    _close(xg.grad, xr.grad, "norm dx", ulp=2.0, abs_frac=4e-3)
    if affine:
        _close(gg.grad, gr.grad, "dgamma", ulp=0.0, abs_frac=2e-3)
        _close(bg.grad, br.grad, "dbeta", ulp=0.0, abs_frac=2e-3)
    if act == "prelu":
        _close(ag.grad, ar.grad, "dalpha", ulp=0.0, abs_frac=2e-3)


def test_pool_upsample_cat_add_bf16():
    ops = _ops()
    torch.manual_seed(3)
    x = _rb(torch.randn(2, 8, 8, 10, 12))
    xr = x.clone().requires_grad_(True)
    pr = F.max_pool3d(xr, 2)
    ur = F.interpolate(pr, scale_factor=2, mode="trilinear", align_corners=False)
    nr = F.interpolate(pr, scale_factor=2, mode="nearest")
    cr = torch.cat((ur, nr), dim=1)
    sr = cr[:, :8] + xr
    dy1, dy2 = _rb(torch.randn_like(cr)), _rb(torch.randn_like(sr))
    (cr * dy1).sum().backward(retain_graph=True)
    g_cat = xr.grad.clone()

    xg = _dev(x).requires_grad_(True)
    pg = ops.max_pool3d(xg, 2)
    ug = ops.upsample3d(pg, scale_factor=2, mode="trilinear", align_corners=False)
    ng = ops.upsample3d(pg, scale_factor=2, mode="nearest")
    cg = ops.cat_channels([ug, ng])
    assert pg.dtype == BF and ug.dtype == BF and cg.dtype == BF
    torch.testing.assert_close(pg.float().cpu(), pr.detach(), rtol=0, atol=0)      # max-pool is exact
    torch.testing.assert_close(ng.float().cpu(), nr.detach(), rtol=0, atol=0)      # nearest is exact
    _close(ug, ur, "trilinear")
    cg.backward(_dev(dy1))
    # bf16 storage of the two intermediate gradients (d ug, d pg) adds up to 2 more roundings
    _close(xg.grad, g_cat, "pool/upsample/cat backward", ulp=4.0, abs_frac=8e-3)
    ag = ops.add(ops.convert(ug.detach(), BF), _dev(x))
    _close(ag, _rb(ur.detach()) + x, "add")


def test_softmax_dice_argmax_bf16():
    ops = _ops()
    torch.manual_seed(9)
    from oracle.losses import softmax_dice_loss
    z = _rb(torch.randn(2, 2, 9, 11, 13) * 2)
    t = (torch.rand(2, 1, 9, 11, 13) < 0.2).float()
    zr = z.clone().requires_grad_(True)
    lr = softmax_dice_loss(zr, t)
    lr.backward()
    zg = _dev(z).requires_grad_(True)
    lg = ops.softmax_dice_loss(zg, t.cuda())
    assert lg.dtype == torch.float32
    lg.backward()
    assert abs(lg.item() - lr.item()) < 1e-5
    _close(zg.grad, zr.grad, "dlogits")
    m = ops.argmax_mask(zg.detach())
    assert torch.equal(m.cpu(), z.argmax(dim=1).to(torch.uint8))


def test_convert_roundtrip_bf16():
    ops = _ops()
    torch.manual_seed(1)
    x = torch.randn(1, 5, 4, 6, 7)
    xb = ops.convert(_dev(x, torch.float32), BF)
    assert xb.dtype == BF and torch.equal(xb.float().cpu(), x.to(BF).float())
    xf = ops.convert(xb, torch.float32)
    assert xf.dtype == torch.float32 and torch.equal(xf.cpu(), x.to(BF).float())
