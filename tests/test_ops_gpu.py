"""GPU parity suite, operator level: every HIP operator (forward AND backward, called through the C ABI via the
autograd seam) against the same op in plain PyTorch fp32 on the CPU — the reference's L2 semantics.
Tolerance: max-norm relative error <= 1e-3 (north_star); index/byte outputs bit-exact."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from mri_epilepsy_diagnosis_amd import ops
from util import assert_close, seeded_rand, seeded_randn, to_ncdhw

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _dev(t, grad=True):
    d = t.to(DEV).contiguous(memory_format=torch.channels_last_3d) if t.dim() == 5 else t.to(DEV)
    return d.requires_grad_(grad) if grad else d


# ---------------------------------------------------------------------------------------------- conv
CONV_CASES = [
    # name, N, Cin, Cout, (D,H,W), k, s, p, d, bias
    ("unet_1_8", 2, 1, 8, (12, 20, 16), 3, 1, 1, 1, True),
    ("unet_8_16", 2, 8, 16, (12, 20, 16), 3, 1, 1, 1, True),
    ("unet_16_16", 1, 16, 16, (16, 12, 20), 3, 1, 1, 1, True),
    ("unet_16_32", 1, 16, 32, (8, 12, 16), 3, 1, 1, 1, True),
    ("unet_32_32", 1, 32, 32, (8, 12, 8), 3, 1, 1, 1, True),
    ("unet_32_64", 1, 32, 64, (8, 6, 8), 3, 1, 1, 1, True),
    ("unet_96_32", 1, 96, 32, (8, 8, 12), 3, 1, 1, 1, True),
    ("unet_48_16", 1, 48, 16, (12, 16, 16), 3, 1, 1, 1, True),
    ("unet_cls_16_2", 2, 16, 2, (12, 20, 16), 1, 1, 0, 1, True),
    ("pw_32_2", 1, 32, 2, (9, 10, 11), 1, 1, 0, 1, False),
    ("pw_64_8", 2, 64, 8, (6, 7, 8), 1, 1, 0, 1, True),
    ("pw_16_5", 1, 16, 5, (6, 7, 8), 1, 1, 0, 1, True),
    ("pw_128_64", 1, 128, 64, (4, 5, 6), 1, 1, 0, 1, False),
    ("pw_8_3", 2, 8, 3, (5, 9, 13), 1, 1, 0, 1, True),       # forward on the lanes-per-voxel kernel: 2 / 1 / 16 lanes per voxel,
    ("pw_4_1", 1, 4, 1, (7, 6, 5), 1, 1, 0, 1, False),       # 1-4 output channels, voxel counts that are no multiple of anything
    ("pw_64_4", 1, 64, 4, (3, 7, 11), 1, 1, 0, 1, True),
    ("mfma_24_40", 1, 24, 40, (6, 9, 17), 3, 1, 1, 1, True),
    ("mfma_64_128_ragged", 1, 64, 128, (3, 5, 7), 3, 1, 1, 1, False),
    ("ragged_3x3x3", 1, 16, 16, (5, 7, 9), 3, 1, 1, 1, False),
    ("tiny_1voxel", 1, 8, 16, (1, 1, 1), 3, 1, 1, 1, True),
    ("sepx_k6s2p2", 2, 1, 8, (32, 12, 10), (6, 1, 1), (2, 1, 1), (2, 0, 0), 1, True),
    ("sepy_k6s2p2", 2, 8, 8, (8, 24, 10), (1, 6, 1), (1, 2, 1), (0, 2, 0), 1, True),
    ("sepz_k6s2p2", 2, 8, 16, (8, 6, 28), (1, 1, 6), (1, 1, 2), (0, 0, 2), 1, True),
    ("sepx_k3p0", 3, 32, 64, (3, 3, 3), (3, 1, 1), 1, 0, 1, True),
    ("sepz_k3p1", 1, 16, 8, (6, 5, 9), (1, 1, 3), 1, (0, 0, 1), 1, True),
    ("stride2_m3d", 1, 8, 16, (12, 10, 14), 3, 2, 1, 1, False),
    ("stride2_odd", 1, 16, 32, (9, 7, 11), 3, 2, 1, 1, False),
    ("dilated_s2", 1, 1, 4, (25, 23, 27), 3, 2, 0, 3, True),
    ("dilated_p3", 1, 4, 4, (11, 12, 13), 3, 1, 3, 3, True),
    ("reduce_k4s4", 1, 1, 1, (16, 12, 8), 4, 4, 0, 1, True),
    ("vox_1_1", 2, 1, 1, (9, 10, 11), 3, 1, 1, 1, True),
    # 1 -> 1 separable convs of the autoencoder's last block (AE_model.py:110-160): the 16-byte stencil kernels (W % 4 == 0) ...
    ("c1_sepy", 2, 1, 1, (9, 10, 12), (1, 3, 1), 1, (0, 1, 0), 1, True),
    ("c1_sepz", 2, 1, 1, (5, 7, 16), (1, 1, 3), 1, (0, 0, 1), 1, True),
    ("c1_sepx_nobias", 1, 1, 1, (6, 5, 8), (3, 1, 1), 1, (1, 0, 0), 1, False),
    ("c1_sepz_k6_p2", 1, 1, 1, (4, 6, 20), (1, 1, 6), 1, (0, 0, 2), 1, True),      # output narrower than the input (W 20 -> 19: gather path)
    ("c1_sepz_k5_p2_dil2", 1, 1, 1, (4, 6, 24), (1, 1, 5), 1, (0, 0, 4), 2, True),  # dilation 2, same width: shifted 16-byte loads
    ("c1_sepy_k2", 2, 1, 1, (4, 9, 8), (1, 2, 1), 1, (0, 1, 0), 1, True),           # even filter: H 9 -> 10
    # ... and a width that is not a multiple of 4 (gather kernels)
    ("c1_sepz_ragged", 1, 1, 1, (5, 6, 10), (1, 1, 3), 1, (0, 0, 1), 1, True),
    ("odd_channels", 1, 3, 5, (6, 7, 8), 3, 1, 1, 1, True),
    ("wide_128", 1, 128, 128, (4, 4, 4), 3, 1, 1, 1, False),
]


@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv3d_fwd_dgrad_wgrad(case):
    _, n, ci, co, sp, k, s, p, d, bias = case
    x = seeded_randn(1, (n, ci, *sp))
    conv = torch.nn.Conv3d(ci, co, k, s, p, d, bias=bias)
    torch.manual_seed(2)
    with torch.no_grad():
        conv.weight.copy_(torch.randn_like(conv.weight) * 0.2)
        if bias:
            conv.bias.copy_(torch.randn_like(conv.bias))
    xr = x.clone().requires_grad_(True)
    yr = conv(xr)
    gy = seeded_randn(3, tuple(yr.shape))
    yr.backward(gy)

    xd = _dev(x)
    w = conv.weight.detach().to(DEV).requires_grad_(True)
    b = conv.bias.detach().to(DEV).requires_grad_(True) if bias else None
    yd = ops.conv3d(xd, w, b, s, p, d)
    assert tuple(yd.shape) == tuple(yr.shape)
    yd.backward(_dev(gy, False))
    assert_close(to_ncdhw(yd), yr, what="y")
    assert_close(to_ncdhw(xd.grad), xr.grad, what="dx")
    assert_close(w.grad.cpu(), conv.weight.grad, what="dw")
    if bias:
        assert_close(b.grad.cpu(), conv.bias.grad, what="db")


def test_conv3d_is_deterministic_and_linear():
    x = _dev(seeded_randn(5, (1, 16, 12, 12, 12)), False)
    w = seeded_randn(6, (16, 16, 3, 3, 3)).to(DEV)
    y1 = ops.conv3d(x, w, None, 1, 1, 1)
    y2 = ops.conv3d(x, w, None, 1, 1, 1)
    assert torch.equal(y1, y2)
    y3 = ops.conv3d(x * 2.0, w, None, 1, 1, 1)
    assert torch.equal(y3, y1 * 2.0)  # exact: scaling by 2 commutes with fp32 rounding


CT_CASES = [("k2s2", 1, 6, 6, (5, 6, 7), 2, 2, 0, 0), ("k4s4_1_1", 2, 1, 1, (3, 4, 5), 4, 4, 0, 0),
            ("k4s2p1", 1, 8, 4, (5, 5, 6), 4, 2, 1, 0), ("k3s2p1op1", 1, 4, 8, (4, 5, 3), 3, 2, 1, 1)]


@pytest.mark.parametrize("case", CT_CASES, ids=[c[0] for c in CT_CASES])
def test_conv_transpose3d(case):
    _, n, ci, co, sp, k, s, p, op = case
    m = torch.nn.ConvTranspose3d(ci, co, k, s, p, op)
    x = seeded_randn(1, (n, ci, *sp))
    xr = x.clone().requires_grad_(True)
    yr = m(xr)
    gy = seeded_randn(2, tuple(yr.shape))
    yr.backward(gy)
    xd = _dev(x)
    w = m.weight.detach().to(DEV).requires_grad_(True)
    b = m.bias.detach().to(DEV).requires_grad_(True)
    yd = ops.conv_transpose3d(xd, w, b, s, p, op, 1)
    yd.backward(_dev(gy, False))
    assert_close(to_ncdhw(yd), yr, what="y")
    assert_close(to_ncdhw(xd.grad), xr.grad, what="dx")
    assert_close(w.grad.cpu(), m.weight.grad, what="dw")
    assert_close(b.grad.cpu(), m.bias.grad, what="db")


# ---------------------------------------------------------------------------------------------- norm + activation
def _ref_act(kind, alpha, slope):
    if kind is None:
        return lambda t: t
    if kind == "relu":
        return F.relu
    if kind == "leaky_relu":
        return lambda t: F.leaky_relu(t, slope)
    return lambda t: F.prelu(t, alpha)


NORM_CASES = [(mode, act, c, an) for mode in ("batch", "instance", "running", "none")
              for act, c, an in ((None, 16, 1), ("relu", 8, 1), ("leaky_relu", 32, 1), ("prelu", 16, 1), ("prelu", 12, 12),
                                 ("prelu", 3, 1))]


@pytest.mark.parametrize("mode,act,c,alpha_n", NORM_CASES)
def test_norm_act_fwd_bwd(mode, act, c, alpha_n):
    if mode == "none" and act is None:
        pytest.skip("identity")
    n, sp = 3, (6, 10, 7)
    x = seeded_randn(1, (n, c, *sp)) * 1.7 + 0.9
    gamma = seeded_randn(2, (c,)) * 0.5 + 1.0
    beta = seeded_randn(3, (c,)) * 0.3
    alpha = (seeded_rand(4, (alpha_n,)) * 0.5 - 0.1)
    rm = seeded_randn(5, (c,)) * 0.1 + 0.8
    rv = seeded_rand(6, (c,)) + 2.0
    affine = mode != "none"

    xr = x.clone().requires_grad_(True)
    gr, br, ar = gamma.clone().requires_grad_(affine), beta.clone().requires_grad_(affine), alpha.clone().requires_grad_(True)
    rm_r, rv_r = rm.clone(), rv.clone()
    if mode == "batch":
        t = F.batch_norm(xr, rm_r, rv_r, gr, br, True, 0.1, 1e-5)
    elif mode == "running":
        t = F.batch_norm(xr, rm_r, rv_r, gr, br, False, 0.1, 1e-5)
    elif mode == "instance":
        t = F.instance_norm(xr, None, None, gr, br, True, 0.1, 1e-5)
    else:
        t = xr
    yr = _ref_act(act, ar, 0.01)(t)
    gy = seeded_randn(7, tuple(yr.shape))
    yr.backward(gy)

    xd = _dev(x)
    gd = gamma.to(DEV).requires_grad_(True) if affine else None
    bd = beta.to(DEV).requires_grad_(True) if affine else None
    ad = alpha.to(DEV).requires_grad_(True) if act == "prelu" else None
    rm_d, rv_d = rm.to(DEV), rv.to(DEV)
    yd = ops.norm_act(xd, gd, bd, ad, rm_d if mode in ("batch", "running") else None,
                      rv_d if mode in ("batch", "running") else None, mode, 0.1, 1e-5, act, 0.01)
    yd.backward(_dev(gy, False))
    assert_close(to_ncdhw(yd), yr, what="y")
    assert_close(to_ncdhw(xd.grad), xr.grad, what="dx")
    if affine:
        assert_close(gd.grad.cpu(), gr.grad, what="dgamma")
        assert_close(bd.grad.cpu(), br.grad, what="dbeta")
    if act == "prelu":
        assert_close(ad.grad.cpu(), ar.grad, what="dalpha")
    if mode == "batch":
        assert_close(rm_d.cpu(), rm_r, rel=1e-5, what="running_mean")
        assert_close(rv_d.cpu(), rv_r, rel=1e-5, what="running_var")


@pytest.mark.parametrize("c,groups,act", [(8, 4, None), (16, 4, "relu"), (12, 3, "prelu"), (6, 1, None), (8, 8, "leaky_relu")])
def test_group_norm_fwd_bwd(c, groups, act):
    x = seeded_randn(1, (3, c, 5, 6, 7)) * 1.3 + 0.4
    gamma, beta = seeded_randn(2, (c,)) * 0.5 + 1.0, seeded_randn(3, (c,)) * 0.3
    alpha = torch.tensor([0.2])
    xr, gr, br, ar = (t.clone().requires_grad_(True) for t in (x, gamma, beta, alpha))
    yr = _ref_act(act, ar, 0.01)(F.group_norm(xr, groups, gr, br, 1e-5))
    gy = seeded_randn(4, tuple(yr.shape))
    yr.backward(gy)
    xd, gd, bd = _dev(x), gamma.to(DEV).requires_grad_(True), beta.to(DEV).requires_grad_(True)
    ad = alpha.to(DEV).requires_grad_(True) if act == "prelu" else None
    yd = ops.norm_act(xd, gd, bd, ad, None, None, "group", 0.1, 1e-5, act, 0.01, None, c // groups)
    yd.backward(_dev(gy, False))
    assert_close(to_ncdhw(yd), yr, what="y")
    assert_close(to_ncdhw(xd.grad), xr.grad, what="dx")
    assert_close(gd.grad.cpu(), gr.grad, what="dgamma")
    assert_close(bd.grad.cpu(), br.grad, what="dbeta")
    if act == "prelu":
        assert_close(ad.grad.cpu(), ar.grad, what="dalpha")


def test_batch_stats_large_offset_is_stable():
    """mean >> std: the shifted-sum statistics must not lose the variance (E[x^2]-E[x]^2 cancellation)."""
    x = seeded_randn(1, (2, 8, 16, 16, 16)) * 0.05 + 300.0
    y = ops.norm_act(_dev(x, False), None, None, None, None, None, "batch", 0.1, 1e-5, None, 0.0)
    ref = F.batch_norm(x, None, None, None, None, True, 0.1, 1e-5)
    assert_close(to_ncdhw(y), ref, rel=5e-3, what="y")


def test_dropout3d_masks_whole_channels():
    x = _dev(seeded_randn(1, (4, 16, 5, 6, 7)))
    torch.manual_seed(0)
    y = ops.dropout3d(x, 0.6, True)
    y.sum().backward()
    yc, xc = to_ncdhw(y), to_ncdhw(x)
    per = (yc != 0).flatten(2).float().mean(-1)
    assert set(per.flatten().tolist()) <= {0.0, 1.0} and 0 < per.mean() < 1
    keep = per.bool()
    assert_close(yc[keep], xc[keep] / 0.4, rel=1e-6)
    assert_close(to_ncdhw(x.grad)[keep], torch.full_like(xc[keep], 2.5), rel=1e-6)
    assert ops.dropout3d(x, 0.6, False) is x


# ---------------------------------------------------------------------------------------------- pooling / upsampling
@pytest.mark.parametrize("c,sp,k,s", [(16, (8, 12, 10), 2, 2), (8, (9, 7, 11), 2, 2), (3, (6, 6, 6), 2, 2),
                                      (4, (13, 12, 14), 4, 2), (1, (8, 8, 8), 2, 2), (64, (4, 6, 4), 2, 2)])
def test_max_pool3d(c, sp, k, s):
    x = seeded_randn(1, (2, c, *sp))
    xr = x.clone().requires_grad_(True)
    yr = F.max_pool3d(xr, k, s)
    gy = seeded_randn(2, tuple(yr.shape))
    yr.backward(gy)
    xd = _dev(x)
    yd = ops.max_pool3d(xd, k, s)
    yd.backward(_dev(gy, False))
    assert torch.equal(to_ncdhw(yd), yr)            # selection: bit-exact
    assert_close(to_ncdhw(xd.grad), xr.grad, rel=1e-6, what="dx")


@pytest.mark.parametrize("c,sp,pad", [(16, (6, 8, 10), 0), (8, (5, 7, 9), 8), (3, (4, 4, 6), 0), (32, (2, 2, 2), 16)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_max_pool3d_skip_sums_both_gradients_in_the_pool_kernel(c, sp, pad, dtype):
    """(pool(x), x) as one node (unet.UNet encoder: `skip = x; x = pool(x)`): outputs equal the two-op form, and the input
    gradient equals maxpool_bwd(dy) + dskip — also when dskip arrives as a pitched channel slice and when one is missing."""
    x = seeded_randn(3, (2, c, *sp))
    if dtype == torch.bfloat16:
        x = x.to(dtype).float()
    xr = x.clone().requires_grad_(True)
    yr = F.max_pool3d(xr, 2)
    gy, gs = seeded_randn(4, tuple(yr.shape)), seeded_randn(5, tuple(x.shape))
    if dtype == torch.bfloat16:
        gy, gs = gy.to(dtype).float(), gs.to(dtype).float()
    (yr * gy).sum().backward(retain_graph=True)
    dpool = xr.grad.clone()
    xd = _dev(x, False).to(dtype).requires_grad_(True)
    yd, skip = ops.max_pool3d_skip(xd, 2)
    assert torch.equal(to_ncdhw(yd).float(), yr.detach()) and skip.data_ptr() == xd.data_ptr() and torch.equal(skip, xd)
    gbuf = torch.zeros(2, c + pad, *sp, device="cuda", dtype=dtype).contiguous(memory_format=torch.channels_last_3d)
    gbuf[:, pad:] = gs.cuda().to(dtype)
    torch.autograd.backward([yd, skip], [_dev(gy, False).to(dtype), gbuf[:, pad:]])
    want = dpool + gs
    tol = 1e-6 if dtype == torch.float32 else 8e-3            # bf16: one rounding of the sum
    assert_close(to_ncdhw(xd.grad).float(), want, rel=tol, what="dx (pool + skip)")
    # only one of the two outputs used
    for use in ("pool", "skip"):
        xd2 = _dev(x, False).to(dtype).requires_grad_(True)
        y2, s2 = ops.max_pool3d_skip(xd2, 2)
        if use == "pool":
            y2.backward(_dev(gy, False).to(dtype))
            assert_close(to_ncdhw(xd2.grad).float(), dpool, rel=tol, what="dx (pool only)")
        else:
            s2.backward(gs.cuda().to(dtype).contiguous(memory_format=torch.channels_last_3d))
            assert_close(to_ncdhw(xd2.grad).float(), gs, rel=tol, what="dx (skip only)")


def test_max_pool3d_ties_pick_first_like_torch():
    x = torch.zeros(1, 4, 4, 4, 4)
    xr = x.clone().requires_grad_(True)
    F.max_pool3d(xr, 2).sum().backward()
    xd = _dev(x)
    ops.max_pool3d(xd, 2).sum().backward()
    assert torch.equal(to_ncdhw(xd.grad), xr.grad)


UP_CASES = [("nearest", None, 2, None, 16, (5, 6, 7)), ("nearest", None, 4, None, 8, (3, 4, 2)),
            ("nearest", (9, 11, 13), None, None, 4, (4, 5, 6)), ("nearest", (7, 7, 7), None, None, 1, (7, 3, 9)),
            ("trilinear", None, 2, False, 32, (5, 6, 4)), ("trilinear", None, 2, True, 8, (5, 6, 4)),
            ("trilinear", (7, 9, 11), None, False, 4, (4, 5, 6)), ("trilinear", None, 2, False, 3, (1, 4, 5)),
            # x2 trilinear marching along D (4 x 16 coarse columns, 16-plane segments): ragged columns in h and w, two segments,
            # 8 / 4 / 2 / 1 pieces per voxel, a half-empty second channel pass (24 channels), two full passes (64)
            ("trilinear", None, 2, False, 64, (18, 9, 19)), ("trilinear", None, 2, False, 24, (3, 5, 33)),
            ("trilinear", None, 2, False, 16, (17, 4, 16)), ("trilinear", None, 2, False, 8, (2, 7, 5)),
            ("trilinear", None, 2, False, 4, (33, 3, 18)), ("trilinear", None, 2, False, 4, (1, 3, 5)), ("trilinear", None, 2, False, 8, (2, 1, 1)),
            ("trilinear", None, 2, False, 12, (5, 1, 21))]


@pytest.mark.parametrize("mode,size,scale,ac,c,sp", UP_CASES)
def test_upsample3d(mode, size, scale, ac, c, sp):
    x = seeded_randn(1, (2, c, *sp))
    xr = x.clone().requires_grad_(True)
    yr = F.interpolate(xr, size=size, scale_factor=scale, mode=mode, align_corners=ac)
    gy = seeded_randn(2, tuple(yr.shape))
    yr.backward(gy)
    xd = _dev(x)
    yd = ops.upsample3d(xd, size=size, scale_factor=scale, mode=mode, align_corners=ac)
    assert tuple(yd.shape) == tuple(yr.shape)
    yd.backward(_dev(gy, False))
    assert_close(to_ncdhw(yd), yr, rel=1e-5, what="y")
    assert_close(to_ncdhw(xd.grad), xr.grad, rel=1e-5, what="dx")


# conv3d over a nearest-upsampled input that is never formed (csrc/upconv.hip; UpBlock of AE_model.py:110-120): the shipped last
# block (8 -> 1, (3,1,1), scale 4), the other two axes, scale 2, a pointwise head, six taps with padding 2, and a shape the fused
# operator does not serve (16 -> 8: falls back to the two operators)
UPCONV_CASES = [(8, 1, (3, 1, 1), (1, 0, 0), 4, (3, 4, 5)), (8, 1, (1, 3, 1), (0, 1, 0), 4, (2, 3, 4)), (4, 2, (1, 1, 3), (0, 0, 1), 2, (5, 6, 7)),
                (8, 8, (1, 1, 1), (0, 0, 0), 4, (2, 2, 3)), (8, 1, (6, 1, 1), (2, 0, 0), 2, (6, 3, 4)), (16, 8, (3, 1, 1), (1, 0, 0), 4, (2, 3, 2)),
                (8, 2, (3, 1, 1), (0, 0, 0), 4, (3, 2, 2))]


@pytest.mark.parametrize("ci,co,k,pad,scale,sp", UPCONV_CASES)
def test_upsample_conv3d_equals_nearest_upsampling_then_conv(ci, co, k, pad, scale, sp):
    x = seeded_randn(1, (2, ci, *sp))
    w = seeded_randn(2, (co, ci, *k)) * 0.3
    b = seeded_randn(3, (co,))
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = F.conv3d(F.interpolate(xr, scale_factor=scale, mode="nearest"), wr, br, padding=pad)
    gy = seeded_randn(4, tuple(yr.shape))
    yr.backward(gy)
    xd = _dev(x)
    wd, bd = w.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
    served = ops.upsample_conv3d_supported(xd, wd, scale, 1, pad, 1)
    assert served == ((ci, co) != (16, 8))
    yd = ops.upsample_conv3d(xd, scale, wd, bd, padding=pad)
    assert tuple(yd.shape) == tuple(yr.shape)
    yd.backward(_dev(gy, False))
    assert_close(to_ncdhw(yd), yr, rel=1e-5, what="y")
    assert_close(to_ncdhw(xd.grad), xr.grad, rel=1e-5, what="dx (coarse)")
    assert_close(wd.grad.cpu(), wr.grad, rel=1e-5, what="dw")
    assert_close(bd.grad.cpu(), br.grad, rel=1e-5, what="db")


def test_upsample_conv3d_is_deterministic_and_matches_the_two_operators_bf16():
    x = seeded_randn(5, (2, 8, 4, 5, 6)).to(torch.bfloat16)
    w = (seeded_randn(6, (1, 8, 3, 1, 1)) * 0.3).to(DEV).requires_grad_(True)
    b = seeded_randn(7, (1,)).to(DEV).requires_grad_(True)
    res = []
    for fused in (True, True, False):
        xd = x.to(DEV).contiguous(memory_format=torch.channels_last_3d).requires_grad_(True)
        w.grad = b.grad = None
        if fused:
            y = ops.upsample_conv3d(xd, 4, w, b, padding=(1, 0, 0))
        else:
            y = ops.conv3d(ops.upsample3d(xd, scale_factor=4, mode="nearest"), w, b, padding=(1, 0, 0))
        g = torch.Generator(device=DEV).manual_seed(9)
        y.backward(torch.randn(y.shape, device=DEV, generator=g).to(y.dtype).contiguous(memory_format=torch.channels_last_3d))
        res.append((y.detach().float(), xd.grad.float(), w.grad.clone(), b.grad.clone()))
    for a, c in zip(res[0], res[1]):
        assert torch.equal(a, c)                      # run to run: bit for bit
    for a, c, tol in zip(res[0], res[2], (2e-2, 3e-2, 1e-3, 1e-3)):   # against the two operators: bf16 storage of the 8-channel tensors
        assert float((a - c).abs().max()) <= tol * float(c.abs().max() + 1e-6)


# the head of the autoencoder's first DownBlock (AE_model.py:45-53): Conv3d(1, 8, (6,1,1), s (2,1,1), p (2,0,0)) then Conv3d(8, 8, (1,k,1),
# s (1,s,1), p (0,p,0)) — the first convolution's weight gradient without the gradient of its output (csrc/sepconv.hip); odd extents,
# a width that is not a multiple of 64, stride 1 and 3 for the second convolution, k = 3
@pytest.mark.parametrize("sp,k2,s2,p2", [((12, 10, 9), 6, 2, 2), ((7, 9, 70), 6, 2, 2), ((8, 6, 5), 3, 1, 1), ((6, 11, 33), 6, 3, 2)])
def test_conv3d_pair_gradients_equal_the_two_convolutions(sp, k2, s2, p2):
    class Conv:   # what ops.conv3d_pair reads of an nn.Conv3d
        def __init__(self, w, b, stride, padding):
            self.weight, self.bias, self.stride, self.padding, self.dilation = w, b, stride, padding, (1, 1, 1)

    x = seeded_randn(1, (2, 1, *sp))
    w1, b1 = seeded_randn(2, (8, 1, 6, 1, 1)) * 0.4, seeded_randn(3, (8,))
    w2, b2 = seeded_randn(4, (8, 8, 1, k2, 1)) * 0.2, seeded_randn(5, (8,))
    ref = [t.clone().requires_grad_(True) for t in (w1, b1, w2, b2)]
    yr = F.conv3d(F.conv3d(x, ref[0], ref[1], stride=(2, 1, 1), padding=(2, 0, 0)), ref[2], ref[3], stride=(1, s2, 1), padding=(0, p2, 0))
    gy = seeded_randn(6, tuple(yr.shape))
    yr.backward(gy)
    dev = [t.to(DEV).requires_grad_(True) for t in (w1, b1, w2, b2)]
    c1, c2 = Conv(dev[0], dev[1], (2, 1, 1), (2, 0, 0)), Conv(dev[2], dev[3], (1, s2, 1), (0, p2, 0))
    xd = _dev(x, False)
    assert ops.conv3d_pair_supported(xd, c1, c2)
    assert not ops.conv3d_pair_supported(_dev(x, True), c1, c2)        # an input that needs its gradient keeps the two operators
    yd = ops.conv3d_pair(xd, c1, c2)
    yd.backward(_dev(gy, False))
    assert_close(to_ncdhw(yd), yr, rel=1e-5, what="y")
    for got, want, what in zip(dev, ref, ("dw1", "db1", "dw2", "db2")):
        assert_close(got.grad.cpu(), want.grad, rel=2e-5, what=what)


# ---------------------------------------------------------------------------------------------- loss / mask / plumbing
@pytest.mark.parametrize("n,c,ct,sp", [(1, 2, 1, (8, 8, 8)), (2, 2, 1, (9, 7, 5)), (2, 3, 3, (6, 6, 6)), (1, 2, 2, (4, 4, 4))])
def test_softmax_dice_loss(n, c, ct, sp):
    from oracle import losses
    lg = seeded_randn(1, (n, c, *sp)) * 2
    tg = (seeded_rand(2, (n, ct, *sp)) < 0.3).float()
    lr = lg.clone().requires_grad_(True)
    loss_r = losses.softmax_dice_loss(lr, tg)
    (loss_r * 1.7).backward()
    ld = _dev(lg)
    loss_d = ops.softmax_dice_loss(ld, _dev(tg, False))
    (loss_d * 1.7).backward()
    assert_close(loss_d.cpu(), loss_r.detach(), rel=1e-5, what="loss")
    assert_close(to_ncdhw(ld.grad), lr.grad, what="dlogits")


def test_softmax_dice_known_answer(golden_dir):
    g = np.load(golden_dir + "/dice_known.npz")
    loss = ops.softmax_dice_loss(_dev(torch.from_numpy(g["logits"]), False), _dev(torch.from_numpy(g["target"]), False))
    np.testing.assert_allclose(loss.item(), float(g["loss"]), rtol=1e-5)


def test_softmax_dice_empty_target_and_all_foreground():
    lg = _dev(seeded_randn(1, (1, 2, 6, 6, 6)), False)
    for fill in (0.0, 1.0):
        tg = torch.full((1, 1, 6, 6, 6), fill)
        from oracle import losses
        ref = losses.softmax_dice_loss(lg.cpu().contiguous(memory_format=torch.contiguous_format), tg)
        assert_close(ops.softmax_dice_loss(lg, _dev(tg, False)).cpu(), ref, rel=1e-5)


def test_argmax_mask_bit_exact():
    lg = seeded_randn(1, (2, 2, 9, 10, 11))
    lg[0, :, 0, 0, 0] = 0.5  # tie -> class 0
    m = ops.argmax_mask(_dev(lg, False))
    assert m.dtype == torch.uint8 and torch.equal(m.cpu(), lg.argmax(dim=1).to(torch.uint8))
    lg3 = seeded_randn(2, (1, 5, 4, 4, 4))
    assert torch.equal(ops.argmax_mask(_dev(lg3, False)).cpu(), lg3.argmax(dim=1).to(torch.uint8))


def test_cat_and_add():
    a, b, c = seeded_randn(1, (2, 16, 4, 5, 6)), seeded_randn(2, (2, 32, 4, 5, 6)), seeded_randn(3, (2, 3, 4, 5, 6))
    ad, bd, cd = _dev(a), _dev(b), _dev(c)
    y = ops.cat_channels([ad, bd, cd])
    gy = seeded_randn(4, tuple(y.shape))
    y.backward(_dev(gy, False))
    assert torch.equal(to_ncdhw(y), torch.cat([a, b, c], 1))
    assert torch.equal(to_ncdhw(ad.grad), gy[:, :16]) and torch.equal(to_ncdhw(bd.grad), gy[:, 16:48])
    assert torch.equal(to_ncdhw(cd.grad), gy[:, 48:])
    a2, b2 = _dev(a), _dev(seeded_randn(5, tuple(a.shape)))
    s = ops.add(a2, b2)
    s.backward(_dev(gy[:, :16].contiguous(), False))
    assert torch.equal(to_ncdhw(s), a + to_ncdhw(b2)) and torch.equal(to_ncdhw(a2.grad), gy[:, :16])


def test_flat_adam_matches_torch_adamw_and_adam():
    from mri_epilepsy_diagnosis_amd import parallel
    for decoupled, ref_cls in ((True, torch.optim.AdamW), (False, torch.optim.Adam)):
        torch.manual_seed(0)
        lin_r = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Linear(5, 3))
        lin_d = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Linear(5, 3))
        lin_d.load_state_dict(lin_r.state_dict())
        lin_d.to(DEV)
        fp = parallel.FlatParams(lin_d)
        opt_d = parallel.FlatAdam(fp, lr=1e-2, weight_decay=0.05, decoupled=decoupled)
        opt_r = ref_cls(lin_r.parameters(), lr=1e-2, weight_decay=0.05)
        for it in range(4):
            x = seeded_randn(10 + it, (6, 7))
            opt_r.zero_grad(); lin_r(x).pow(2).mean().backward(); opt_r.step()
            opt_d.zero_grad(); lin_d(x.to(DEV)).pow(2).mean().backward(); opt_d.step(fp.all_reduce())
        for pr, pd in zip(lin_r.parameters(), lin_d.parameters()):
            assert_close(pd.detach().cpu(), pr.detach(), rel=1e-5)


def test_unsupported_and_invalid_arguments_raise():
    x = _dev(seeded_randn(1, (1, 4, 4, 4, 4)), False)
    with pytest.raises(RuntimeError, match="input channels"):
        ops.conv3d(x, torch.randn(2, 3, 3, 3, 3, device=DEV))
    with pytest.raises(RuntimeError, match="greater than actual input size"):
        ops.conv3d(x, torch.randn(2, 4, 5, 5, 5, device=DEV))     # reference behaviour for the 160x192x160 head
    with pytest.raises(RuntimeError):
        ops.conv3d(x.half(), torch.randn(2, 4, 3, 3, 3, device=DEV).half())
    with pytest.raises(RuntimeError):
        ops.softmax_dice_loss(x, _dev(torch.zeros(1, 2, 4, 4, 4), False))


def test_mask_overlap_counts_dice_iou_exact():
    """§8(f1): Dice / IoU of the validation loop from integer overlap counts taken on the device — bit-exact against the
    oracle (and through it the reference's compute_dice_coefficient / get_iou_score golden values)."""
    import numpy as np
    from oracle import metrics as O_MET
    from util import load_golden
    g = load_golden("mask_metrics.npz")
    for row, pr, d, i in zip(g["cases"], g["probs"], g["dice"], g["iou"]):
        gt, pred = O_MET.seeded_masks(int(row[0]), tuple(int(v) for v in row[1:]), *[float(v) for v in pr])
        c = ops.mask_overlap_counts(torch.from_numpy(pred).cuda(), torch.from_numpy(gt).cuda())
        assert c.tolist() == [int(gt.sum()), int(pred.sum()), int((gt & pred).sum()),
                              int(np.logical_and(gt > 0, pred > 0).sum()), int(np.logical_or(gt > 0, pred > 0).sum())]
        dsc, iou = ops.dice_iou_from_counts(c)
        assert dsc == d and iou == i
    # general uint8 values (label ids), unaligned views, ragged size
    rng = np.random.Generator(np.random.PCG64(5))
    gt = rng.integers(0, 256, size=100003, dtype=np.uint8)
    pred = rng.integers(0, 4, size=100003, dtype=np.uint8)
    c = ops.mask_overlap_counts(torch.from_numpy(pred).cuda()[3:], torch.from_numpy(gt).cuda()[3:])
    gt, pred = gt[3:], pred[3:]
    assert c.tolist() == [int(gt.sum(dtype=np.int64)), int(pred.sum(dtype=np.int64)), int((gt & pred).sum(dtype=np.int64)),
                          int(np.logical_and(gt > 0, pred > 0).sum()), int(np.logical_or(gt > 0, pred > 0).sum())]
    # full-size volume (BASELINE shape) against numpy
    gt, pred = O_MET.seeded_masks(77, (160, 192, 160), 0.1, 0.1, 0.9)
    dsc, iou = ops.dice_iou_from_counts(ops.mask_overlap_counts(torch.from_numpy(pred).cuda(), torch.from_numpy(gt).cuda()))
    assert dsc == O_MET.dice_coefficient(gt, pred) and iou == O_MET.iou_score(pred, gt)
    z = torch.zeros(4, 4, 4, dtype=torch.uint8, device="cuda")
    assert ops.mask_overlap_counts(z, z).tolist() == [0, 0, 0, 0, 0]    # both empty: Dice is NaN in the reference


def test_surface_distance_device_vs_oracle_and_reference_golden():
    """§8(f4): average surface distances from the on-device neighbour codes + exact Euclidean distance transform against the
    oracle (scipy) and, through tests/golden/surface_asd.npz, the reference's own compute_average_surface_distance.  The
    squared distances are exact integers; only the order of the two float64 sums differs (1e-12 relative)."""
    import numpy as np
    from mri_epilepsy_diagnosis_amd.segmentation import surface
    from oracle import metrics as O_MET
    from util import load_golden
    g = load_golden("surface_asd.npz")
    area = g["area_table"]
    for row, ref in zip(g["cases"], g["asd"]):
        gt, pred = O_MET.seeded_blobs(int(row[0]), tuple(int(v) for v in row[1:]))
        got = surface.average_surface_distance(torch.from_numpy(gt).cuda(), torch.from_numpy(pred).cuda())
        assert np.allclose(got, ref, rtol=1e-12, atol=0), (got, ref)
    a = np.zeros((32, 32, 32), np.uint8); a[4:20, 4:20, 4:20] = 1
    b = np.zeros((32, 32, 32), np.uint8); b[6:22, 4:20, 4:20] = 1
    got = surface.average_surface_distance(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda())
    assert np.allclose(got, g["cube_asd"], rtol=1e-12) and abs(got[0] - 0.671674) < 1e-6
    # edge cases of the reference: one mask empty -> inf / nan, both empty -> nan, nan; mask touching the volume border
    z = np.zeros((8, 9, 10), np.uint8)
    full = np.ones((8, 9, 10), np.uint8)
    for gt, pred in ((a[:8, :9, :10], z), (z, z), (full, a[:8, :9, :10] | 1), (full, z)):
        got = surface.average_surface_distance(torch.from_numpy(np.ascontiguousarray(gt)).cuda(),
                                               torch.from_numpy(np.ascontiguousarray(pred)).cuda())
        ref = O_MET.average_surface_distance(gt, pred, area)
        assert np.allclose(got, ref, rtol=1e-12, atol=0, equal_nan=True), (got, ref)
    # full-size volume
    gt, pred = O_MET.seeded_blobs(5, (160, 192, 160))
    got = surface.average_surface_distance(torch.from_numpy(gt).cuda(), torch.from_numpy(pred).cuda())
    assert np.allclose(got, O_MET.average_surface_distance(gt, pred, area), rtol=1e-12, atol=0)


def test_surface_element_lists_and_order_metrics_bit_exact():
    """§8(f4): the device's surface-element lists (exact squared distances + neighbour codes), sorted the reference's way, must
    EQUAL the oracle's lists element for element; robust Hausdorff 95, surface overlap and surface Dice at 1 mm computed from
    them are then bit-identical to the reference's values (tests/golden/surface_asd.npz), SURVEY's known answers included."""
    import numpy as np
    from mri_epilepsy_diagnosis_amd.segmentation import surface
    from oracle import metrics as O_MET
    from util import load_golden
    g = load_golden("surface_asd.npz")
    area = g["area_table"]
    for row, hd, sdc, ov, asd in zip(g["cases"], g["hd95"], g["sdice1"], g["overlap1"], g["asd"]):
        gt, pred = O_MET.seeded_blobs(int(row[0]), tuple(int(v) for v in row[1:]))
        sd = surface.surface_distances(torch.from_numpy(gt).cuda(), torch.from_numpy(pred).cuda())
        ref = O_MET.surface_distances(gt, pred, area)
        for key in ref:
            assert np.array_equal(sd[key], ref[key]), key
        assert surface.compute_robust_hausdorff(sd, 95) == hd
        assert surface.compute_surface_dice_at_tolerance(sd, 1) == sdc
        assert np.array_equal(surface.compute_surface_overlap_at_tolerance(sd, 1), ov)
        assert np.array_equal(surface.compute_average_surface_distance(sd), asd)          # bit-exact through the sorted lists
    a = np.zeros((32, 32, 32), np.uint8); a[4:20, 4:20, 4:20] = 1
    b = np.zeros((32, 32, 32), np.uint8); b[6:22, 4:20, 4:20] = 1
    sd = surface.surface_distances(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda())
    assert surface.compute_robust_hausdorff(sd, 95) == 2.0
    assert surface.compute_surface_dice_at_tolerance(sd, 1) == float(g["cube_sdice1"])
    # one empty mask: inf distances; both empty: empty lists
    z = np.zeros((6, 7, 8), np.uint8)
    sd = surface.surface_distances(torch.from_numpy(np.ascontiguousarray(a[:6, :7, :8] | 1)).cuda(), torch.from_numpy(z).cuda())
    assert len(sd["distances_pred_to_gt"]) == 0 and np.isinf(sd["distances_gt_to_pred"]).all()
    assert surface.compute_robust_hausdorff(sd, 95) == np.inf
    sd = surface.surface_distances(torch.from_numpy(z).cuda(), torch.from_numpy(z).cuda())
    assert all(len(v) == 0 for v in sd.values())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("case", [(2, 16, 16, 9, 13, 21, True), (1, 48, 16, 8, 16, 32, True), (2, 8, 32, 5, 8, 16, False),
                                  (1, 32, 64, 4, 9, 17, True), (1, 16, 48, 6, 7, 19, True),
                                  (64, 16, 16, 9, 13, 21, True), (90, 48, 16, 3, 9, 17, False), (88, 16, 48, 5, 7, 19, True)],
                         ids=lambda c: "n%d_%d-%d_%dx%dx%d_b%d" % c)
def test_batchnorm_statistics_from_the_conv_epilogue_equal_the_statistics_pass(case, dtype):
    """conv3d(bn_stats=True) accumulates the BatchNorm batch statistics in the MFMA kernel's epilogue (float64 partials per
    workgroup, shift = bias); norm_act must then give what it gives with its own statistics pass over y: outputs, running
    statistics, and all gradients — on ragged sizes (masked tile borders), one and two N-tiles per wave, several passes (Cout 48)."""
    nb, ci, co, d, h, w, has_bias = case
    g = torch.Generator().manual_seed(7)
    x = (torch.randn(nb, ci, d, h, w, generator=g) * 1.5 + 0.3).cuda().to(dtype).contiguous(memory_format=torch.channels_last_3d)
    wt = (torch.randn(co, ci, 3, 3, 3, generator=g) / np.sqrt(27 * ci)).cuda()
    b = (torch.randn(co, generator=g) * 2.0).cuda() if has_bias else None
    gamma, beta = (torch.rand(co, generator=g) + 0.5).cuda(), torch.randn(co, generator=g).cuda()
    dy = torch.randn(nb, co, d, h, w, generator=g).cuda().to(dtype).contiguous(memory_format=torch.channels_last_3d)
    res = []
    for fused in (False, True):
        xs = x.clone().requires_grad_(True)
        ws, bs = wt.clone().requires_grad_(True), (b.clone().requires_grad_(True) if has_bias else None)
        gs, be = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
        rm, rv = torch.zeros(co, device="cuda"), torch.ones(co, device="cuda")
        y = ops.conv3d(xs, ws, bs, padding=1, bn_stats=fused)
        assert hasattr(y, "_mri3d_bn_stats") == fused
        z = ops.norm_act(y, gs, be, None, rm, rv, "batch", 0.1, 1e-5, "relu")
        z.backward(dy)
        res.append([z.detach().float(), rm, rv, xs.grad.float(), ws.grad, gs.grad, be.grad] + ([bs.grad] if has_bias else []))
    tol = 2e-5 if dtype == torch.float32 else 2e-2     # bf16: the fused statistics see y before its rounding to bf16
    names = ["z", "running_mean", "running_var", "dx", "dw", "dgamma", "dbeta"] + (["dbias"] if has_bias else [])
    for name, a, r in zip(names, res[1], res[0]):
        scale = r.abs().max().item()
        if name == "dbias":      # analytically zero (a bias in front of a batch-statistics BatchNorm): both are rounding noise
            scale = res[0][4].abs().max().item()
        # bf16 gradients: a statistic that moves in its 5th digit flips ReLU decisions of near-zero pre-activations, and each flip
        # moves individual dx / dw entries by one bf16-rounded term: the more voxels, the larger the largest single such move, so
        # those are held to a relative L2 bound (the flips are sparse) and a loose max bound
        if dtype == torch.bfloat16 and name in ("dx", "dw", "dgamma", "dbeta", "dbias"):
            assert (a - r).norm().item() <= 2e-2 * (r.norm().item() + scale), (name, (a - r).norm().item(), r.norm().item())
            assert (a - r).abs().max().item() <= 0.2 * (scale + 1e-6), (name, (a - r).abs().max().item(), scale)
            continue
        assert (a - r).abs().max().item() <= tol * (scale + 1e-6), (name, (a - r).abs().max().item(), scale)
    # against torch's own batch statistics (fp32 only: exact semantics check incl. the unbiased running variance)
    if dtype == torch.float32:
        yr = F.conv3d(x.float().cpu(), wt.cpu(), b.cpu() if has_bias else None, padding=1)
        m, v = yr.mean(dim=(0, 2, 3, 4)), yr.var(dim=(0, 2, 3, 4), unbiased=True)
        assert torch.allclose(res[1][1].cpu(), 0.1 * m, rtol=1e-4, atol=1e-5)
        assert torch.allclose(res[1][2].cpu(), 0.9 + 0.1 * v, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_max_pool3d_2_nan_and_index_semantics_equal_torch(dtype):
    """MaxPool3d(2) on even extents (the eight-loads-at-once kernel): NaNs propagate, the FIRST maximum of a window wins, and the
    gradient lands on torch's arg-max voxel — on a tensor with repeated values and NaNs sprinkled in."""
    g = torch.Generator().manual_seed(12)
    x = torch.randint(-3, 4, (2, 16, 6, 8, 10), generator=g).float()       # many ties
    x[torch.rand(x.shape, generator=g) < 0.02] = float("nan")
    xr = x.clone().requires_grad_(True)
    yr = F.max_pool3d(xr, 2)
    gy = torch.randn(yr.shape, generator=g)
    if dtype == torch.bfloat16:
        gy = gy.to(dtype).float()
    yr.backward(gy)
    xd = _dev(x, False).to(dtype).requires_grad_(True)
    yd = ops.max_pool3d(xd, 2)
    yd.backward(_dev(gy, False).to(dtype))
    a, r = to_ncdhw(yd).float().cpu(), yr.detach()
    assert torch.equal(torch.isnan(a), torch.isnan(r)) and torch.equal(a[~torch.isnan(r)], r[~torch.isnan(r)])
    assert torch.equal(to_ncdhw(xd.grad).float().cpu(), xr.grad)


# (batch, Ca, Cb, Cout, volume, channel padding of the second tensor's buffer, bias, served by the split kernels in fp32 / bf16)
CAT_CASES = [(2, 16, 32, 16, (24, 40, 70), 0, True, True, True), (2, 32, 64, 32, (21, 33, 70), 8, True, True, True),
             (8, 32, 64, 32, (45, 17, 37), 8, True, True, True),       # bf16: the weight gradient marches along d (3 segments of 20)
             (5, 16, 16, 8, (17, 40, 65), 0, False, True, True),       # (>= 512 work units in every pass: below that the plain convolution
                                                                       #  prefers the LDS-free kernel and the sums are ordered differently)
             (2, 16, 24, 16, (24, 40, 70), 0, True, False, True),      # 24 trailing channels: not a ci-tile multiple for the fp32 weight gradient
             (1, 8, 16, 16, (24, 40, 70), 0, True, False, False),      # 8 leading channels: the split must be a multiple of 16
             (1, 16, 32, 16, (6, 7, 9), 0, True, False, True)]         # tiny volume: fp32 runs on the LDS-free kernel (no split support)


def test_conv3d_cat_between_256_and_512_work_units_stays_on_the_tiled_kernel():
    """256 <= work units < 512 (fp32): the plain convolution prefers the LDS-free MFMA kernel, the split-operand one keeps the
    tiled kernel — different summation orders, so the two agree to rounding, not bit for bit; the split path must still be served."""
    g = torch.Generator().manual_seed(5)
    sp = (17, 40, 65)
    xa = torch.randn(3, 16, *sp, generator=g).cuda().contiguous(memory_format=torch.channels_last_3d)
    xb = torch.randn(3, 16, *sp, generator=g).cuda().contiguous(memory_format=torch.channels_last_3d)
    wt = (torch.randn(8, 32, 3, 3, 3, generator=g) / np.sqrt(27 * 32)).cuda()
    dy = torch.randn(3, 8, *sp, generator=g).cuda().contiguous(memory_format=torch.channels_last_3d)
    res = []
    for split in (True, False):
        a, b, w = xa.clone().requires_grad_(True), xb.clone().requires_grad_(True), wt.clone().requires_grad_(True)
        if split:
            y = ops.conv3d_cat(a, b, w, None, padding=1)
            assert "Conv3dCatFn" in type(y.grad_fn).__name__
        else:
            y = ops.conv3d(torch.cat((a, b), dim=1).contiguous(memory_format=torch.channels_last_3d), w, None, padding=1)
        y.backward(dy)
        res.append((y.detach(), a.grad, b.grad, w.grad))
    for name, p, q in zip(("y", "dxa", "dxb", "dw"), res[0], res[1]):
        scale = q.abs().max().item()
        assert (p - q).abs().max().item() <= 2e-5 * scale, (name, (p - q).abs().max().item(), scale)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("case", CAT_CASES, ids=lambda c: "n%d_%d+%d-%d_%s_p%d_b%d" % (c[0], c[1], c[2], c[3], "x".join(map(str, c[4])), c[5], c[6]))
def test_conv3d_cat_equals_conv_of_the_concatenation(case, dtype):
    """ops.conv3d_cat(xa, xb, w) — the U-Net decoder's conv over cat((skip, upsampled)) read from the two tensors — against
    ops.conv3d on the explicit concatenation: same kernels, same summation order, so outputs, both input gradients, weight and bias
    gradients are BIT-identical; with the second tensor as a pitched channel slice; with fused BatchNorm statistics; and on shapes
    the split kernels do not serve (8 leading channels, tiny volume, 24 trailing channels in fp32), where it must fall back to
    the concatenation."""
    nb, ca, cb, co, sp, pad_b, has_bias, served_f32, served_bf16 = case
    served = served_f32 if dtype == torch.float32 else served_bf16
    g = torch.Generator().manual_seed(ca * 100 + cb)
    xa = torch.randn(nb, ca, *sp, generator=g).cuda().to(dtype).contiguous(memory_format=torch.channels_last_3d)
    bbuf = torch.zeros(nb, cb + pad_b, *sp, device="cuda", dtype=dtype).contiguous(memory_format=torch.channels_last_3d)
    bbuf[:, pad_b:] = torch.randn(nb, cb, *sp, generator=g).cuda().to(dtype)
    wt = (torch.randn(co, ca + cb, 3, 3, 3, generator=g) / np.sqrt(27 * (ca + cb))).cuda()
    b = torch.randn(co, generator=g).cuda() if has_bias else None
    dy = torch.randn(nb, co, *sp, generator=g).cuda().to(dtype).contiguous(memory_format=torch.channels_last_3d)
    res = []
    for split in (False, True):
        a = xa.clone().requires_grad_(True)
        bb = bbuf[:, pad_b:].detach().requires_grad_(True) if split else bbuf[:, pad_b:].clone().requires_grad_(True)
        w, bs = wt.clone().requires_grad_(True), (b.clone().requires_grad_(True) if has_bias else None)
        for stats in (False, True):
            if split:
                y = ops.conv3d_cat(a, bb, w, bs, padding=1, bn_stats=stats)
                assert ("Conv3dCatFn" in type(y.grad_fn).__name__) == served, (type(y.grad_fn).__name__, served)
            else:
                y = ops.conv3d(torch.cat((a, bb), dim=1).contiguous(memory_format=torch.channels_last_3d), w, bs, padding=1, bn_stats=stats)
            if stats:
                st = getattr(y, "_mri3d_bn_stats", None)
                res.append((y.detach().clone(), None if st is None else st[0].view(st[1], -1).sum(0).clone()))
        y.backward(dy)
        res.append((a.grad, bb.grad, w.grad, None if bs is None else bs.grad))
    (y0, s0), g0, (y1, s1), g1 = res
    assert torch.equal(y0, y1)
    assert (s0 is None) == (s1 is None) and (s0 is None or torch.allclose(s0, s1, rtol=1e-12, atol=0))
    for name, p, q in zip(("dxa", "dxb", "dw", "db"), g0, g1):
        assert (p is None) == (q is None)
        if p is not None:
            if name in ("dxa", "dxb"):
                assert torch.equal(p, q), name
            else:   # the ci-tile -> workgroup map is the same, so is the order of the sums
                assert torch.equal(p, q), name
