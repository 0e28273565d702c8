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
    (1, 32, 2, (9, 10, 11), 1, 1, 0, 1),        # pointwise heads: 8 channels per lane, vector dy loads; ragged voxel counts
    (2, 64, 4, (5, 7, 9), 1, 1, 0, 1),
    (1, 16, 5, (6, 7, 8), 1, 1, 0, 1),
    (1, 24, 3, (6, 7, 8), 1, 1, 0, 1),
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


@pytest.mark.parametrize("c,sp", [(8, (5, 9, 19)), (16, (17, 4, 16)), (32, (3, 6, 33)), (64, (18, 5, 7)), (12, (4, 5, 18)), (40, (2, 9, 17))])
def test_trilinear_x2_forward_backward_bf16(c, sp):
    """nn.Upsample(scale_factor=2, mode='trilinear') of the U-Net decoder in the bf16 region: the marching kernels with 16-byte lane
    items (channels in octets) and with 8-byte ones (12 channels), ragged columns, two D segments, one and two channel passes."""
    ops = _ops()
    torch.manual_seed(c)
    x = _rb(torch.randn(2, c, *sp))
    xr = x.clone().requires_grad_(True)
    ur = F.interpolate(xr, scale_factor=2, mode="trilinear", align_corners=False)
    dy = _rb(torch.randn_like(ur))
    ur.backward(dy)
    xg = _dev(x).requires_grad_(True)
    ug = ops.upsample3d(xg, scale_factor=2, mode="trilinear", align_corners=False)
    assert ug.dtype == BF
    _close(ug, ur, "trilinear x2 forward")
    ug.backward(_dev(dy))
    _close(xg.grad, xr.grad, "trilinear x2 backward")


@pytest.mark.parametrize("scale", [2, 4])
def test_nearest_upsampling_backward_with_an_integer_scale_bf16(scale):
    """nn.Upsample(scale_factor=S, mode='nearest') backward on bf16 tensors (the S^3 box-sum kernel; AE_model.py:110-120 uses S = 4):
    fp32 accumulation of the S^3 fine gradients, one rounding of the result."""
    ops = _ops()
    torch.manual_seed(11)
    x = _rb(torch.randn(2, 8, 3, 5, 7))
    xr = x.clone().requires_grad_(True)
    yr = F.interpolate(xr, scale_factor=scale, mode="nearest")
    dy = _rb(torch.randn_like(yr))
    yr.backward(dy)
    xg = _dev(x).requires_grad_(True)
    yg = ops.upsample3d(xg, scale_factor=scale, mode="nearest")
    torch.testing.assert_close(yg.float().cpu(), yr.detach(), rtol=0, atol=0)
    yg.backward(_dev(dy))
    _close(xg.grad, xr.grad, "nearest x%d backward" % scale, ulp=1.0, abs_frac=1e-3)


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


# ------------------------------------------------------------------------------------------------ model level
def _unet(c0=8):
    from mri_epilepsy_diagnosis_amd.unet import UNet
    return UNet(in_channels=1, out_classes=2, dimensions=3, num_encoding_blocks=3, out_channels_first_layer=c0,
                normalization="batch", upsampling_type="linear", padding=True, activation="PReLU")


def _cos(a, b):
    a, b = a.detach().double().flatten().cpu(), b.detach().double().flatten().cpu()
    return (a @ b / (a.norm() * b.norm() + 1e-300)).item()


def test_unet_bf16_autocast_step_tracks_fp32_oracle():
    """BASELINE configs[3] semantics: bf16 activations, fp32 parameters / statistics / loss / parameter gradients.
    bf16 carries 8 significand bits (eps 2^-8 = 3.9e-3), so the comparison with the fp32 oracle is statistical:
    loss within 1e-2, logits within 5e-2 of their range, every parameter gradient pointing the same way (cosine)."""
    ops = _ops()
    from oracle import losses, unet_recon
    torch.manual_seed(21)
    orc = unet_recon.UNetRecon(out_channels_first_layer=8)
    prod = _unet(8)
    prod.load_state_dict(orc.state_dict())
    prod.to("cuda")
    x = torch.randn(2, 1, 32, 48, 32)
    t = (torch.rand(2, 1, 32, 48, 32) < 0.1).float()
    orc.train()
    out_o = orc(x)
    loss_o = losses.softmax_dice_loss(out_o, t)
    loss_o.backward()
    prod.train()
    with ops.autocast():
        out_p = prod(x.cuda())
        loss_p = ops.softmax_dice_loss(out_p, t.cuda())
    assert out_p.dtype == BF and loss_p.dtype == torch.float32
    loss_p.backward()
    assert abs(loss_p.item() - loss_o.item()) <= 1e-2 * abs(loss_o.item())
    scale = out_o.detach().abs().max().item()
    assert (out_p.float().cpu() - out_o.detach()).abs().max().item() <= 5e-2 * scale
    gmax = max(po.grad.abs().max().item() for po in orc.parameters())
    for (name, po), (_, pp) in zip(orc.named_parameters(), prod.named_parameters()):
        assert pp.grad is not None and pp.grad.dtype == torch.float32 and pp.dtype == torch.float32, name
        if po.numel() < 16:
            # PReLU slopes: one number summed over every voxel with both signs — compare on the scale of the sum's terms
            assert (pp.grad.cpu() - po.grad).abs().max().item() <= 0.5 * po.grad.abs().max().item() + 1e-2 * gmax, name
            continue
        if po.grad.norm().item() < 1e-7:
            continue
        c = _cos(pp.grad, po.grad)
        assert c > 0.95, "%s: gradient cosine %.4f" % (name, c)
        r = pp.grad.norm().item() / po.grad.norm().item()
        assert 0.8 < r < 1.25, "%s: gradient norm ratio %.3f" % (name, r)
    # BatchNorm running statistics are updated from fp32 statistics of the bf16 activations
    for (name, bo), (_, bp) in zip(orc.named_buffers(), prod.named_buffers()):
        if bo.dtype.is_floating_point:
            assert bp.dtype == torch.float32
            assert (bp.cpu() - bo).abs().max().item() <= 2e-2 * (bo.abs().max().item() + 1e-3), name


def test_unet_bf16_training_reduces_loss_like_fp32():
    """Five AdamW steps on a fixed batch: the bf16 region must follow the fp32 HIP run's loss curve."""
    ops = _ops()
    from mri_epilepsy_diagnosis_amd import parallel
    torch.manual_seed(4)
    x = torch.randn(2, 1, 32, 32, 32).cuda()
    t = (torch.rand(2, 1, 32, 32, 32) < 0.15).float().cuda()
    curves = {}
    for mode in ("fp32", "bf16"):
        torch.manual_seed(7)
        net = _unet(8).cuda()
        flat = parallel.FlatParams(net)
        opt = parallel.FlatAdam(flat, lr=1e-3)
        ls = []
        for _ in range(5):
            flat.zero_grad()
            with ops.autocast(enabled=(mode == "bf16")):
                loss = ops.softmax_dice_loss(net(x), t)
            loss.backward()
            opt.step(flat.all_reduce())
            ls.append(loss.item())
        curves[mode] = ls
    assert curves["bf16"][-1] < curves["bf16"][0]
    for a, b in zip(curves["fp32"], curves["bf16"]):
        assert abs(a - b) <= 2e-2 * abs(a), curves


def test_unet_bf16_eval_mask_agrees_with_fp32():
    """Inference: arg-max masks from the bf16 region agree with fp32 except where the two logits are within bf16
    resolution of each other."""
    ops = _ops()
    torch.manual_seed(2)
    net = _unet(8).cuda().eval()
    x = torch.randn(1, 1, 32, 48, 32).cuda()
    with torch.no_grad():
        z32 = net(x)
        with ops.autocast():
            z16 = net(x)
    m32, m16 = ops.argmax_mask(z32), ops.argmax_mask(z16)
    margin = (z32[:, 0] - z32[:, 1]).abs()
    differ = (m32 != m16)
    scale = z32.abs().max().item()
    assert differ.float().mean().item() < 0.05
    assert (margin[differ] <= 8e-2 * scale).all()
