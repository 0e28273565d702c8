"""The d-marching forward / data-gradient kernel (csrc/conv_march.hip) BY NAME through the C ABI
(mri3d_conv3d_fwd_march / mri3d_conv3d_dgrad_march), against torch's CPU convolution in fp32 on the same inputs:
nn.Conv3d(k=3, stride=1, padding=1) forward and its data gradient (unet.UNet's ConvolutionalBlock convs,
/root/reference segmentation/routine.py:346-356).  fp32: north_star's 1e-3 relative (max-norm); bf16: fp32 arithmetic on the
SAME bf16-rounded inputs / weights up to the final rounding of the stored result (2 bf16 ulp + 2e-3 of the scale).
Covers one column, ragged h / w, several workgroup shapes, several d segments, 1..4 channel chunks, half-filled chunks,
1..3 output blocks, the two-tensor input (conv over cat((x, x2))) and two-tensor data gradient, fused BatchNorm statistics."""
import ctypes

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

BF = torch.bfloat16
CL = torch.channels_last_3d


def _env():
    from mri_epilepsy_diagnosis_amd import _lib, ops
    return _lib, ops


def _rb(t):
    return t.to(BF).float()


def _dev(t, dtype):
    return t.to("cuda").to(dtype).contiguous(memory_format=CL)


def _close(got, ref, what, dtype):
    got = got.detach().float().cpu()
    ref = ref.detach().float().cpu()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    scale = ref.abs().max().item() + 1e-30
    err = (got - ref).abs()
    if dtype == BF:
        tol = 2.0 * 2.0 ** -8 * ref.abs() + 2e-3 * scale
        bad = err > tol
        assert not bad.any(), "%s: %d/%d outside tolerance, max err %.3e (scale %.3e)" % (what, int(bad.sum()), bad.numel(), err.max().item(), scale)
    else:
        assert err.max().item() <= 1e-3 * scale, "%s: max-norm relative error %.3e" % (what, err.max().item() / scale)


def _march_fwd(x, w, b, x2=None, stats=False):
    """y (and the [blocks][co][2] float64 statistics partials) of conv(cat((x, x2)), w) + b through mri3d_conv3d_fwd_march."""
    _lib, ops = _env()
    L = _lib.lib()
    n, c1, d, h, wd = x.shape
    ci = c1 + (x2.shape[1] if x2 is not None else 0)
    co = w.shape[0]
    g = ops._conv_geom((n, ci, d, h, wd), w.shape, (1, 1, 1), (1, 1, 1), (1, 1, 1), x_ld=c1, y_ld=co, dtype=ops._dt(x))
    gq = ops._conv_geom((n, ci, d, h, wd), w.shape, (1, 1, 1), (1, 1, 1), (1, 1, 1), dtype=ops._dt(x))   # (the query takes one tensor)
    assert L.mri3d_conv3d_march_supported(ctypes.byref(gq), _lib.PASS_FWD) == 1
    y = torch.empty((n, co, d, h, wd), dtype=x.dtype, device=x.device, memory_format=CL)
    ws = ops._workspace(L.mri3d_conv3d_workspace_bytes(ctypes.byref(g), _lib.PASS_FWD), x.device)
    part = None
    if stats:
        nb = L.mri3d_conv3d_march_stats_blocks(ctypes.byref(gq))
        assert nb > 0
        part = torch.full((nb, co, 2), float("nan"), dtype=torch.float64, device=x.device)
    _lib.check(L.mri3d_conv3d_fwd_march(ctypes.byref(g), ops._ptr(x), ops._ptr(x2), c1, 0 if x2 is None else x2.shape[1],
                                        ops._ptr(w), ops._ptr(b), ops._ptr(y), ops._ptr(part), ops._ptr(ws), ws.numel(),
                                        ops._stream()), "conv3d_fwd_march")
    return y, part


def _march_dgrad(dy, w, split=None):
    _lib, ops = _env()
    L = _lib.lib()
    n, co, d, h, wd = dy.shape
    ci = w.shape[1]
    c1 = ci if split is None else split
    g = ops._conv_geom((n, ci, d, h, wd), w.shape, (1, 1, 1), (1, 1, 1), (1, 1, 1), x_ld=c1, y_ld=co, dtype=ops._dt(dy))
    gq = ops._conv_geom((n, ci, d, h, wd), w.shape, (1, 1, 1), (1, 1, 1), (1, 1, 1), dtype=ops._dt(dy))
    assert L.mri3d_conv3d_march_supported(ctypes.byref(gq), _lib.PASS_DGRAD) == 1
    dx = torch.empty((n, c1, d, h, wd), dtype=dy.dtype, device=dy.device, memory_format=CL)
    dx2 = None if split is None else torch.empty((n, ci - c1, d, h, wd), dtype=dy.dtype, device=dy.device, memory_format=CL)
    ws = ops._workspace(L.mri3d_conv3d_workspace_bytes(ctypes.byref(g), _lib.PASS_DGRAD), dy.device)
    _lib.check(L.mri3d_conv3d_dgrad_march(ctypes.byref(g), ops._ptr(dy), ops._ptr(w), ops._ptr(dx), ops._ptr(dx2), c1,
                                          0 if split is None else ci - c1, ops._ptr(ws), ws.numel(), ops._stream()),
               "conv3d_dgrad_march")
    return dx, dx2


def _supported(ci, co, size, n, dtype, which):
    _lib, ops = _env()
    g = ops._conv_geom((n, ci) + tuple(size), (co, ci, 3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1), dtype=_lib.BF16 if dtype == BF else _lib.F32)
    return _lib.lib().mri3d_conv3d_march_supported(ctypes.byref(g), _lib.PASS_FWD if which == "fwd" else _lib.PASS_DGRAD)


CASES = [
    # (n, ci, co, (d, h, w))
    (1, 16, 16, (5, 8, 16)),       # exactly one column, one segment
    (2, 16, 16, (9, 11, 21)),      # ragged rows and voxels, two samples
    (1, 8, 16, (6, 9, 17)),        # bf16: half-filled chunk
    (1, 32, 16, (7, 16, 32)),      # bf16 two chunks / fp32 four; 2 x 2 columns
    (1, 16, 48, (6, 10, 20)),      # three output blocks
    (1, 16, 16, (26, 8, 16)),      # several d segments
    (1, 24, 8, (10, 24, 16)),      # 8 output channels; bf16 second chunk half filled; 3 x 1 columns
    (1, 16, 32, (4, 70, 8)),       # tall and narrow: an 8 x 1 workgroup shape, w < 16
    (3, 16, 16, (3, 5, 130)),      # wide: a 1 x 8 workgroup shape with a ragged ninth column; depth 3
    (1, 16, 16, (1, 8, 16)),       # a single plane
]


@pytest.mark.parametrize("dtype", [torch.float32, BF], ids=["f32", "bf16"])
@pytest.mark.parametrize("case", CASES, ids=lambda c: "n%d_%d-%d_%s" % (c[0], c[1], c[2], "x".join(map(str, c[3]))))
def test_march_forward_and_data_gradient_vs_torch_cpu(case, dtype):
    n, ci, co, size = case
    torch.manual_seed(sum(size) + ci + co)
    rnd = _rb if dtype == BF else (lambda t: t)
    x = rnd(torch.randn(n, ci, *size))
    conv = torch.nn.Conv3d(ci, co, 3, padding=1)
    w, b = conv.weight.detach(), conv.bias.detach()
    wq = rnd(w)      # the kernel rounds the weights to the tensor dtype for the MFMA operands
    xr = x.clone().requires_grad_(True)
    yr = F.conv3d(xr, wq, b, padding=1)
    dy = rnd(torch.randn_like(yr))
    yr.backward(dy)

    # the kernel keeps the weights of one 16-channel output block in LDS: at most four 32-byte chunks of input channels
    ck = 16 if dtype == BF else 8
    if -(-ci // ck) <= 4:
        y, _ = _march_fwd(_dev(x, dtype), w.cuda(), b.cuda())
        _close(y, yr, "forward", dtype)
    else:
        assert _supported(ci, co, size, n, dtype, "fwd") == 0
    if -(-co // ck) <= 4:
        dx, _ = _march_dgrad(_dev(dy, dtype), w.cuda())
        _close(dx, xr.grad, "data gradient", dtype)
    else:
        assert _supported(ci, co, size, n, dtype, "dgrad") == 0


@pytest.mark.parametrize("dtype", [torch.float32, BF], ids=["f32", "bf16"])
def test_march_split_operands_and_fused_statistics(dtype):
    """conv over cat((skip, upsampled)) read from two tensors, its data gradient written to two tensors, and the BatchNorm batch
    statistics of the result (sum a, sum a^2 of a = y - bias in float64 partials) — unet.UNet's decoder conv."""
    torch.manual_seed(5)
    rnd = _rb if dtype == BF else (lambda t: t)
    c1, c2, co, size = 16, 16 if dtype == torch.float32 else 32, 16, (11, 13, 37)
    xa, xb = rnd(torch.randn(2, c1, *size)), rnd(torch.randn(2, c2, *size))
    conv = torch.nn.Conv3d(c1 + c2, co, 3, padding=1)
    w, b = conv.weight.detach(), conv.bias.detach()
    wq = rnd(w)
    xr = torch.cat((xa, xb), 1).requires_grad_(True)
    ar = F.conv3d(xr, wq, None, padding=1)
    yr = ar + b.view(1, -1, 1, 1, 1)
    dy = rnd(torch.randn_like(yr))
    yr.backward(dy)

    y, part = _march_fwd(_dev(xa, dtype), w.cuda(), b.cuda(), x2=_dev(xb, dtype), stats=True)
    _close(y, yr, "forward over two tensors", dtype)
    sums = part.sum(0).cpu()
    assert torch.isfinite(sums).all()
    ref = torch.stack((ar.double().sum((0, 2, 3, 4)), (ar.double() ** 2).sum((0, 2, 3, 4))), 1)
    # the statistics are taken from the fp32 accumulators, before the result is rounded for storage
    rel = ((sums - ref).abs() / (ref.abs() + 1e-3 * ref.abs().max())).max().item()
    assert rel < (2e-3 if dtype == BF else 1e-4), rel
    dxa, dxb = _march_dgrad(_dev(dy, dtype), w.cuda(), split=c1)
    _close(dxa, xr.grad[:, :c1], "data gradient, first tensor", dtype)
    _close(dxb, xr.grad[:, c1:], "data gradient, second tensor", dtype)


def test_march_statistics_equal_the_tiled_kernels_statistics_full_plane():
    """Same layer through the dispatcher's tiled kernel (fp32 small volume: never the marching kernel) and the marching kernel:
    outputs equal to fp32 rounding, statistics equal to 1e-6."""
    _lib, ops = _env()
    torch.manual_seed(9)
    x = _dev(torch.randn(1, 16, 12, 24, 40), torch.float32)
    w, b = torch.randn(16, 16, 3, 3, 3, device="cuda") * 0.1, torch.randn(16, device="cuda")
    y_m, part = _march_fwd(x, w, b, stats=True)
    y_t = ops.conv3d(x, w, b, padding=1)
    assert ((y_m - y_t).abs().max() / y_t.abs().max()).item() < 1e-5
    a = (y_t - b.view(1, -1, 1, 1, 1)).double()
    ref = torch.stack((a.sum((0, 2, 3, 4)), (a * a).sum((0, 2, 3, 4))), 1)
    assert ((part.sum(0) - ref).abs() / ref.abs().max()).max().item() < 1e-6


def test_march_refuses_what_it_cannot_compute():
    _lib, ops = _env()
    L = _lib.lib()
    # fp32 with more than four 8-channel chunks: the weights of an output block would not fit beside the plane buffers
    g = ops._conv_geom((1, 48, 8, 8, 16), (16, 48, 3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1), dtype=_lib.F32)
    assert L.mri3d_conv3d_march_supported(ctypes.byref(g), _lib.PASS_FWD) == 0
    # strided
    g = ops._conv_geom((1, 16, 8, 8, 16), (16, 16, 3, 3, 3), (2, 2, 2), (1, 1, 1), (1, 1, 1), dtype=_lib.F32)
    assert L.mri3d_conv3d_march_supported(ctypes.byref(g), _lib.PASS_FWD) == 0
    x = torch.zeros(1, 48, 8, 8, 16, device="cuda").contiguous(memory_format=CL)
    with pytest.raises(_lib.Mri3dError):
        _march_fwd_unchecked(x, torch.zeros(16, 48, 3, 3, 3, device="cuda"))


def _march_fwd_unchecked(x, w):
    _lib, ops = _env()
    L = _lib.lib()
    n, ci, d, h, wd = x.shape
    g = ops._conv_geom((n, ci, d, h, wd), w.shape, (1, 1, 1), (1, 1, 1), (1, 1, 1), dtype=ops._dt(x))
    y = torch.empty((n, w.shape[0], d, h, wd), dtype=x.dtype, device=x.device, memory_format=CL)
    ws = ops._workspace(1 << 20, x.device)
    _lib.check(L.mri3d_conv3d_fwd_march(ctypes.byref(g), ops._ptr(x), None, 0, 0, ops._ptr(w), None, ops._ptr(y), None,
                                        ops._ptr(ws), ws.numel(), ops._stream()), "conv3d_fwd_march")
