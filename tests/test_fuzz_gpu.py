"""Seeded random-geometry sweep of the 3x3x3 MFMA conv kernels (fp32 and bf16 storage): ragged spatial sizes around the tile
edges (4x8x16 forward tile, 2x6x16 / 2x6x32 weight-gradient tiles), every channel count the dispatcher routes to an MFMA
kernel, pitched (channel-slice) inputs and outputs — against torch's CPU conv on the same (rounded) inputs."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from mri_epilepsy_diagnosis_amd import ops

pytestmark = pytest.mark.gpu


def _cases(n, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    chans = [8, 16, 24, 32, 48, 64]
    out = []
    for _ in range(n):
        ci, co = int(rng.choice(chans)), int(rng.choice(chans))
        d, h, w = int(rng.integers(1, 11)), int(rng.integers(1, 15)), int(rng.integers(1, 40))
        nb = int(rng.integers(1, 3))
        pad_in, pad_out = int(rng.choice([0, 8, 16])), int(rng.choice([0, 8]))
        out.append((nb, ci, co, d, h, w, pad_in, pad_out, int(rng.integers(0, 1 << 30))))
    return out


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("case", _cases(14, 2024), ids=lambda c: "n%d_%d-%d_%dx%dx%d_p%d_%d" % c[:8])
def test_conv3x3x3_random_geometry(case, dtype):
    _run_conv_case(case, dtype)


# channel slices whose base address is NOT 16-byte aligned (pad_in = 2 fp32 elements / 2 or 4 bf16 elements, odd pitches):
# legal inputs that the MFMA kernels (16-byte pieces) cannot take — the dispatcher must fall back to the generic kernels
# instead of failing with EINVAL
MISALIGNED = [(1, 16, 16, 5, 9, 20, 2, 0, 11), (2, 8, 16, 4, 8, 16, 2, 2, 12), (1, 48, 16, 3, 8, 17, 4, 4, 13),
              (1, 16, 32, 6, 10, 18, 6, 0, 14), (1, 32, 32, 2, 3, 5, 1, 3, 15)]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("case", MISALIGNED, ids=lambda c: "n%d_%d-%d_%dx%dx%d_p%d_%d" % c[:8])
def test_conv3x3x3_misaligned_channel_slices_fall_back_to_generic_kernels(case, dtype):
    _run_conv_case(case, dtype)


# fp32 volumes with fewer than 256 work units go to the wave-per-M-tile kernel (conv_mfma_small_kernel); these ragged shapes
# have enough batch to stay on the TILED fp32 kernel, so that its border tiles keep their fp32 coverage
TILED_F32 = [(40, 16, 16, 5, 9, 19, 0, 0, 21), (24, 8, 32, 6, 11, 17, 8, 0, 22), (48, 48, 16, 3, 7, 21, 0, 8, 23)]


@pytest.mark.parametrize("case", TILED_F32, ids=lambda c: "n%d_%d-%d_%dx%dx%d_p%d_%d" % c[:8])
def test_conv3x3x3_ragged_tiles_with_large_batch_stay_on_the_tiled_kernel(case):
    _run_conv_case(case, torch.float32)


# deep-level shapes of Modified3DUNet (batch 1): served by the small-volume kernel in fp32 (1, 2 and 4 N-tiles per wave)
SMALL = [(1, 64, 64, 20, 24, 20, 0, 0, 31), (1, 128, 128, 10, 12, 10, 0, 0, 32), (1, 128, 64, 20, 24, 20, 0, 0, 33),
         (1, 32, 32, 40, 48, 40, 0, 0, 34), (2, 24, 48, 7, 5, 9, 8, 8, 35), (1, 16, 16, 3, 3, 3, 0, 0, 36)]


@pytest.mark.parametrize("case", SMALL, ids=lambda c: "n%d_%d-%d_%dx%dx%d_p%d_%d" % c[:8])
def test_conv3x3x3_small_volumes(case):
    _run_conv_case(case, torch.float32)


# one-N-tile passes with >= 256 (tile, N-block) units take the 64-byte-chunk kernel (conv_mfma3.hip, 8x8x16 tiles): ragged tile
# borders in every dimension, 1 / 2 / 3 chunks (fp32 16 channels, bf16 32 channels per chunk; bf16 48 = a half-empty second chunk),
# three N-blocks (the 48-channel data gradient), pitched inputs and outputs
LARGE_BATCH_CASES = [(64, 16, 16, 9, 13, 21, 0, 0, 41), (72, 48, 16, 5, 9, 19, 0, 8, 42), (96, 16, 48, 3, 10, 17, 16, 0, 43),
            (40, 32, 16, 11, 9, 33, 8, 8, 44), (260, 16, 16, 2, 3, 5, 0, 0, 45),
            # volumes at most 8 voxels wide go to the LDS-free MFMA kernel whatever the batch (two rows per 16-voxel M-tile; an M-tile
            # may straddle rows, planes and samples): the patch CNN's 8^3 level, a ragged one, pitched, > 32 MB of input
            (128, 32, 64, 8, 8, 8, 0, 0, 46), (70, 16, 24, 5, 7, 6, 8, 0, 47), (300, 64, 64, 8, 8, 8, 0, 0, 48)]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("case", LARGE_BATCH_CASES, ids=lambda c: "n%d_%d-%d_%dx%dx%d_p%d_%d" % c[:8])
def test_conv3x3x3_many_small_volumes(case, dtype):
    _run_conv_case(case, dtype)


# bf16 weight gradient marching along d (conv_mfma_wgrad_bf16t_kernel, LDS-DMA rows + transposing reads: taken when columns x segments >= 3 tasks per workgroup): a last
# segment of 5 / 3 / 1 planes, a last row tile of one row, a last column tile of 5 voxels, an 8-channel input tile (upper half
# empty) read from a pitched buffer, 24 output channels (half-empty second block) written from a pitched gradient
MARCH_CASES = [(8, 64, 64, 45, 17, 37, 0, 0, 91), (24, 8, 128, 23, 9, 33, 8, 0, 92), (64, 32, 24, 41, 12, 20, 0, 8, 93)]


@pytest.mark.parametrize("case", MARCH_CASES, ids=lambda c: "n%d_%d-%d_%dx%dx%d_p%d_%d" % c[:8])
def test_bf16_weight_gradient_marching_along_d(case):
    _run_conv_case(case, torch.bfloat16)


# exactly 8 output channels in the forward (8 -> 8, 16 -> 8, 24 -> 8) or in the data gradient (8 -> 16, 8 -> 32): the row-paired
# variant of the tiled MFMA kernel (two taps share the 16-row weight operand, 9 accumulators, halves folded in the epilogue);
# big enough for the tiled path (>= 256 tiles) and ragged in every axis; pitched slices; one case with a single chunk
N8_CASES = [(2, 8, 8, 21, 35, 50, 0, 0, 71), (1, 16, 8, 17, 40, 65, 8, 8, 72), (2, 24, 8, 9, 33, 47, 0, 0, 73),
            (1, 8, 16, 13, 41, 70, 0, 8, 74), (1, 8, 32, 9, 34, 49, 8, 0, 75), (40, 8, 8, 4, 8, 16, 0, 0, 76),
            # weight gradient with paired operand halves (v6, CI8 / CO8): 32 -> 8, pitched 8 -> 8, one-tile and W < 16 volumes
            (1, 32, 8, 11, 19, 37, 0, 0, 77), (1, 8, 8, 7, 13, 33, 8, 8, 78), (3, 8, 8, 2, 6, 16, 0, 0, 79), (2, 16, 8, 5, 7, 9, 0, 0, 80)]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("case", N8_CASES, ids=lambda c: "n%d_%d-%d_%dx%dx%d_p%d_%d" % c[:8])
def test_conv3x3x3_eight_output_channels(case, dtype):
    _run_conv_case(case, dtype)


# stride-2 3x3x3 layers (modified_3dunet.py:23-38, cnn_model.py:49-81) on the LDS-free MFMA kernel: forward over output M-tiles,
# data gradient over same-parity input M-tiles with wave-uniform tap sets; even / odd extents (the last output voxel then has no
# kw = 2 neighbour), pitched slices, few units (a workgroup per unit, taps split over its waves) and many, Kc = 8 (half a chunk)
STRIDED = [(1, 8, 16, 16, 18, 20, 0, 0, 51), (1, 16, 32, 9, 11, 13, 0, 0, 52), (2, 32, 64, 10, 12, 9, 8, 16, 53),
           (1, 64, 128, 6, 7, 5, 0, 0, 54), (1, 8, 16, 40, 48, 40, 0, 0, 55), (3, 24, 40, 5, 6, 33, 0, 8, 56),
           (1, 16, 8, 7, 9, 37, 0, 0, 57), (1, 8, 16, 2, 3, 1, 0, 0, 58)]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("case", STRIDED, ids=lambda c: "n%d_%d-%d_%dx%dx%d_p%d_%d" % c[:8])
def test_conv3x3x3_stride2(case, dtype):
    _run_conv_case(case, dtype, stride=2)


@pytest.mark.parametrize("case", [(1, 16, 32, 10, 11, 13, 0, 0, 61), (2, 8, 16, 7, 8, 19, 8, 0, 62)], ids=lambda c: "n%d_%d-%d_%dx%dx%d_p%d_%d" % c[:8])
def test_conv3x3x3_stride3(case):
    _run_conv_case(case, torch.float32, stride=3)


def _run_conv_case(case, dtype, stride=1):
    nb, ci, co, d, h, w, pad_in, pad_out, seed = case
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(nb, ci, d, h, w, generator=g)
    wt = torch.randn(co, ci, 3, 3, 3, generator=g) * (1.0 / np.sqrt(27 * ci))
    b = torch.randn(co, generator=g)
    do, ho, wo = [(e - 1) // stride + 1 for e in (d, h, w)]     # k 3, pad 1
    dy = torch.randn(nb, co, do, ho, wo, generator=g)
    if dtype == torch.bfloat16:
        x, dy = x.to(dtype).float(), dy.to(dtype).float()
    # input as a channel slice of a wider NDHWC buffer (voxel pitch ci + pad_in), as the decoder's concat buffers are
    xbuf = torch.zeros(nb, ci + pad_in, d, h, w, device="cuda", dtype=dtype).contiguous(memory_format=torch.channels_last_3d)
    xbuf[:, pad_in:] = x.cuda().to(dtype)
    xg = xbuf[:, pad_in:].detach().requires_grad_(True)
    wg, bg = wt.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    yg = ops.conv3d(xg, wg, bg, stride=stride, padding=1)
    dybuf = torch.zeros(nb, co + pad_out, do, ho, wo, device="cuda", dtype=dtype).contiguous(memory_format=torch.channels_last_3d)
    dybuf[:, :co] = dy.cuda().to(dtype)
    yg.backward(dybuf[:, :co])

    def ref(wref):
        xr = x.clone().requires_grad_(True)
        wr, br = wref.clone().requires_grad_(True), b.clone().requires_grad_(True)
        yr = F.conv3d(xr, wr, br, stride=stride, padding=1)
        yr.backward(dy)
        return yr.detach(), xr.grad, wr.grad, br.grad

    if dtype == torch.float32:
        yr, dxr, dwr, dbr = ref(wt)
        tol_act = tol_par = 2e-5
    else:
        # forward / data-gradient round the weights to bf16 for the MFMA operands; the weight gradient does not involve them
        yr, dxr, _, _ = ref(wt.to(dtype).float())
        _, _, dwr, dbr = ref(wt)
        tol_act, tol_par = 1.2e-2, 2e-3

    def close(a, r, tol, what):
        a, r = a.detach().float().cpu(), r.float()
        err = (a - r).abs().max().item()
        assert err <= tol * (r.abs().max().item() + 1e-6), "%s: %.3e vs scale %.3e" % (what, err, r.abs().max().item())

    close(yg, yr, tol_act, "y")
    close(xg.grad, dxr, tol_act, "dx")
    close(wg.grad, dwr, tol_par, "dw")
    close(bg.grad, dbr, tol_par, "db")


FIRST = [(2, 8, 9, 13, 37, True, 0), (1, 16, 8, 16, 32, False, 0), (1, 8, 4, 8, 32, False, 8), (3, 16, 5, 7, 19, True, 0),
         (1, 8, 1, 1, 1, True, 0), (1, 8, 17, 9, 70, True, 8), (2, 16, 6, 20, 33, True, 16), (1, 8, 12, 24, 64, False, 0)]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("case", FIRST, ids=lambda c: "n%d_1-%d_%dx%dx%d_b%d_p%d" % c)
def test_first_layer_conv_one_input_channel(case, dtype):
    """Conv3d(1, 8|16, 3, padding=1): the direct first-layer kernels (conv_cin1_{fwd,wgrad}_kernel), ragged tiles, outputs
    written into / gradients read from a pitched channel slice."""
    nb, co, d, h, w, bias, pad_out = case
    g = torch.Generator().manual_seed(co * 1000 + d * 100 + h * 10 + w)
    x = torch.randn(nb, 1, d, h, w, generator=g)
    wt = torch.randn(co, 1, 3, 3, 3, generator=g) * 0.2
    b = torch.randn(co, generator=g) if bias else None
    dy = torch.randn(nb, co, d, h, w, generator=g)
    if dtype == torch.bfloat16:
        x, dy = x.to(dtype).float(), dy.to(dtype).float()
    xg = x.cuda().to(dtype).requires_grad_(True)
    wg = wt.cuda().requires_grad_(True)
    bg = b.cuda().requires_grad_(True) if bias else None
    yg = ops.conv3d(xg, wg, bg, padding=1)
    dybuf = torch.zeros(nb, co + pad_out, d, h, w, device="cuda", dtype=dtype).contiguous(memory_format=torch.channels_last_3d)
    dybuf[:, pad_out:] = dy.cuda().to(dtype)
    yg.backward(dybuf[:, pad_out:])
    x64, w64 = x.double().requires_grad_(True), wt.double().requires_grad_(True)
    b64 = b.double().requires_grad_(True) if bias else None
    r = F.conv3d(x64, w64, b64, padding=1)
    r.backward(dy.double())
    tol_act = 2e-5 if dtype == torch.float32 else 1.2e-2       # bf16: y / dx are rounded to bf16 on store
    tol_par = 2e-5 if dtype == torch.float32 else 2e-5         # parameter gradients accumulate in fp32 from exact inputs

    def close(a, ref, tol, what):
        err = (a.detach().double().cpu() - ref).abs().max().item()
        assert err <= tol * (ref.abs().max().item() + 1e-6), "%s: %.3e vs scale %.3e" % (what, err, ref.abs().max().item())

    close(yg, r.detach(), tol_act, "y")
    close(xg.grad, x64.grad, tol_act, "dx")
    close(wg.grad, w64.grad, tol_par, "dw")
    if bias:
        close(bg.grad, b64.grad, tol_par, "db")


ONE_OUT = [(2, 8, (3, 1, 1), (1, 0, 0), (1, 1, 1), (9, 13, 37), True), (1, 1, (1, 3, 1), (0, 1, 0), (1, 1, 1), (4, 8, 32), False),
           (1, 1, (1, 1, 3), (0, 0, 1), (1, 1, 1), (5, 7, 19), True), (2, 4, (1, 6, 1), (0, 2, 0), (1, 2, 1), (6, 20, 9), True),
           (1, 16, (1, 1, 3), (0, 0, 1), (1, 1, 1), (3, 5, 70), False), (2, 1, (3, 3, 3), (1, 1, 1), (1, 1, 1), (9, 13, 37), True),
           (1, 1, (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1), True), (1, 8, (6, 1, 1), (2, 0, 0), (2, 1, 1), (12, 6, 10), True)]


@pytest.mark.parametrize("case", ONE_OUT, ids=lambda c: "n%d_%d-1_k%s_s%s_%s" % (c[0], c[1], "x".join(map(str, c[2])),
                                                                                   "".join(map(str, c[4])), "x".join(map(str, c[5]))))
def test_single_output_channel_convs(case):
    """The autoencoder's single-channel tail (AE_model.py:110-160): few-tap convs ending in one channel (co1 weight-gradient
    kernel) and the 1 -> 1 3x3x3 `vox` stencil (forward, data gradient, weight gradient)."""
    nb, ci, k, pad, stride, shape, bias = case
    g = torch.Generator().manual_seed(ci * 100 + sum(shape))
    x = torch.randn((nb, ci) + shape, generator=g)
    wt = torch.randn((1, ci) + k, generator=g) * 0.3
    b = torch.randn(1, generator=g) if bias else None
    xg = x.cuda().contiguous(memory_format=torch.channels_last_3d).requires_grad_(True)
    wg = wt.cuda().requires_grad_(True)
    bg = b.cuda().requires_grad_(True) if bias else None
    yg = ops.conv3d(xg, wg, bg, stride=stride, padding=pad)
    x64, w64 = x.double().requires_grad_(True), wt.double().requires_grad_(True)
    b64 = b.double().requires_grad_(True) if bias else None
    r = F.conv3d(x64, w64, b64, stride=stride, padding=pad)
    dy = torch.randn(r.shape, generator=g)
    yg.backward(dy.cuda())
    r.backward(dy.double())

    def close(a, ref, what):
        err = (a.detach().double().cpu() - ref).abs().max().item()
        assert err <= 2e-5 * (ref.abs().max().item() + 1e-6), "%s: %.3e vs scale %.3e" % (what, err, ref.abs().max().item())

    close(yg, r.detach(), "y")
    close(xg.grad, x64.grad, "dx")
    close(wg.grad, w64.grad, "dw")
    if bias:
        close(bg.grad, b64.grad, "db")


def test_conv_tensors_with_more_than_2_31_elements():
    """288 GB of HBM invite batches whose activation tensors exceed 2^31 ELEMENTS (12 x 48 x 160x192x160 = 2.83e9, 11.3 GB in fp32):
    every voxel index in the kernels must be 64-bit (or relative to a per-item origin).  Size-independent property: the kernels are
    deterministic per output voxel, so the last volume of the big batch must equal — bit for bit — the same volume convolved alone
    (forward and data gradient), and the weight gradient of the batch must equal the sum of the per-volume weight gradients."""
    nb, ci, co, shape = 12, 48, 16, (160, 192, 160)
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.empty((nb, ci) + shape, device="cuda", memory_format=torch.channels_last_3d)
    assert x.numel() > 2 ** 31
    for i in range(nb):                       # generated volume by volume (a single randn of 11 GB is its own stress test)
        x[i].copy_(torch.randn((ci,) + shape, device="cuda", generator=g).unsqueeze(0).contiguous(memory_format=torch.channels_last_3d)[0])
    wt = (torch.randn(co, ci, 3, 3, 3, device="cuda", generator=g) / np.sqrt(27 * ci))
    b = torch.randn(co, device="cuda", generator=g)
    dy = torch.randn((nb, co) + shape, device="cuda", generator=g).contiguous(memory_format=torch.channels_last_3d)

    def run(xs, dys):
        xs = xs.detach().requires_grad_(True)
        w, bb = wt.clone().requires_grad_(True), b.clone().requires_grad_(True)
        y = ops.conv3d(xs, w, bb, padding=1)
        y.backward(dys)
        return y.detach(), xs.grad, w.grad, bb.grad

    y_all, dx_all, dw_all, db_all = run(x, dy)
    dw_sum, db_sum = torch.zeros_like(dw_all, dtype=torch.float64), torch.zeros_like(db_all, dtype=torch.float64)
    for i in (0, nb // 2, nb - 1):
        y_i, dx_i, _, _ = run(x[i:i + 1], dy[i:i + 1])
        assert torch.equal(y_all[i:i + 1], y_i), "forward of volume %d differs inside the big batch" % i
        assert torch.equal(dx_all[i:i + 1], dx_i), "data gradient of volume %d differs inside the big batch" % i
    del y_all, dx_all
    for i in range(nb):
        _, _, dw_i, db_i = run(x[i:i + 1], dy[i:i + 1])
        dw_sum += dw_i.double()
        db_sum += db_i.double()
    assert torch.allclose(dw_all.double(), dw_sum, rtol=2e-5, atol=2e-5 * float(dw_sum.abs().max()))
    assert torch.allclose(db_all.double(), db_sum, rtol=2e-5, atol=2e-5 * float(db_sum.abs().max()))


def test_streaming_ops_on_tensors_with_more_than_2_31_elements():
    """Same property for the per-sample streaming operators (InstanceNorm + LeakyReLU, MaxPool3d, trilinear x2 on the pooled
    tensor) on a 2.83e9-element activation tensor: volume i of the batch result equals the operator on volume i alone, forward
    and backward, bit for bit."""
    nb, c, shape = 12, 48, (160, 192, 160)
    g = torch.Generator(device="cuda").manual_seed(6)
    x = torch.empty((nb, c) + shape, device="cuda", memory_format=torch.channels_last_3d)
    for i in range(nb):
        x[i].copy_(torch.randn((c,) + shape, device="cuda", generator=g).unsqueeze(0).contiguous(memory_format=torch.channels_last_3d)[0])
    assert x.numel() > 2 ** 31

    def run(xs):
        xs = xs.detach().requires_grad_(True)
        z = ops.norm_act(xs, None, None, None, None, None, "instance", 0.1, 1e-5, "leaky_relu", 0.01)
        p = ops.max_pool3d(z, 2)
        u = ops.upsample3d(p, scale_factor=2, mode="trilinear", align_corners=False)
        (u * u).sum().backward()
        return z.detach(), p.detach(), u.detach(), xs.grad

    big = run(x)
    for i in (0, nb - 1):
        one = run(x[i:i + 1])
        for name, a, r in zip(("norm_act", "max_pool", "upsample", "dx"), big, one):
            assert torch.equal(a[i:i + 1], r), "%s of volume %d differs inside the big batch" % (name, i)
