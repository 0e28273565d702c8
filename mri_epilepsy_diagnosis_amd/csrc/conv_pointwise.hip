// conv_pointwise.hip — 1x1x1 Conv3d data- and weight-gradient for narrow heads at full resolution: the 16->2
// classifier of unet.UNet (segmentation/routine.py:346-356 builds it; SURVEY Appendix A.1 `classifier`) and the
// deep-supervision / localisation 1x1x1 convs of modified_3dunet.py:42-65.  Both passes are pure HBM streams
// (algorithmic bytes: one read of x and dy, one write of dx); the direct generic kernels spend their time on
// uncoalesced 4-byte stores and LDS staging instead, so these get their own lane mapping:
//   dgrad: lane = (voxel, 4-channel quad of dx)   -> every wave stores 1 KiB contiguous
//   wgrad: lane = (voxel lane, 4-channel quad of x) with a 4 x CO register tile, block-reduced through LDS in double,
//          one partial per block, fixed-order final sum (deterministic).
#include "common.h"

namespace mri3d {

// CO = compile-time bound on Co (weights of the lane's channel quad live in registers); QC = Ci/4 must divide 256 so
// that a lane's quad never changes across the grid-stride loop; 4 independent voxels per iteration keep loads in flight.
template <typename T, int CO>
__global__ void __launch_bounds__(256)
pw_dgrad_kernel(const T* __restrict__ dy, const float* __restrict__ w, const float* __restrict__ bias,
                T* __restrict__ dx, int64_t nvox, int Ci, int Co, int x_ld, int y_ld) {
    const int QC = Ci >> 2;
    const int q = threadIdx.x % QC, vl = threadIdx.x / QC, VL = 256 / QC;
    float4 wq[CO];
#pragma unroll
    for (int co = 0; co < CO; ++co) {   // scalar loads: parameters may be views into a flat buffer (only 4-byte aligned)
        const float* wr = w + (size_t)(co < Co ? co : 0) * Ci + 4 * q;
        wq[co] = co < Co ? make_float4(wr[0], wr[1], wr[2], wr[3]) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const float4 b4 = bias ? make_float4(bias[4 * q], bias[4 * q + 1], bias[4 * q + 2], bias[4 * q + 3])
                           : make_float4(0.f, 0.f, 0.f, 0.f);
    constexpr int U = 4;
    const int64_t stride = (int64_t)gridDim.x * VL;
    for (int64_t v0 = (int64_t)blockIdx.x * VL + vl; v0 < nvox; v0 += stride * U) {
        float gv[U][CO];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t v = v0 + u * stride;
#pragma unroll
            for (int co = 0; co < CO; ++co) gv[u][co] = (v < nvox && co < Co) ? ldf(dy + v * y_ld + co) : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t v = v0 + u * stride;
            if (v >= nvox) break;
            float4 acc = b4;
#pragma unroll
            for (int co = 0; co < CO; ++co) {
                acc.x = fmaf(gv[u][co], wq[co].x, acc.x);
                acc.y = fmaf(gv[u][co], wq[co].y, acc.y);
                acc.z = fmaf(gv[u][co], wq[co].z, acc.z);
                acc.w = fmaf(gv[u][co], wq[co].w, acc.w);
            }
            stf4(dx + v * x_ld + 4 * q, acc);
        }
    }
}

// Forward of the same narrow heads (16 -> 2 at full resolution): lane = (voxel, 4-channel quad of x) — a wave reads 1 KiB
// contiguous per load — the QC = Ci/4 lanes of a voxel are neighbours and sum their CO partial dot products with QC - 1 lane
// exchanges each (QC a power of two <= 16), lane q == 0 stores the voxel's CO outputs.  The generic few-tap kernel gave every
// voxel to ONE lane (four 16-byte loads at a 64-byte lane pitch): 3.5 TB/s on the U-Net's classifier.
template <typename T, int CO>
__global__ void __launch_bounds__(256)
pw_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias, T* __restrict__ y,
              int64_t nvox, int Ci, int Co, int x_ld, int y_ld) {
    const int QC = Ci >> 2;
    const int q = threadIdx.x % QC, vl = threadIdx.x / QC, VL = 256 / QC;
    float4 wq[CO];
    float bq[CO];
#pragma unroll
    for (int co = 0; co < CO; ++co) {   // scalar loads: parameters may be views into a flat buffer (only 4-byte aligned)
        const float* wr = w + (size_t)(co < Co ? co : 0) * Ci + 4 * q;
        wq[co] = co < Co ? make_float4(wr[0], wr[1], wr[2], wr[3]) : make_float4(0.f, 0.f, 0.f, 0.f);
        bq[co] = (bias != nullptr && co < Co) ? bias[co] : 0.f;
    }
    constexpr int U = 4;
    const int64_t stride = (int64_t)gridDim.x * VL;
    for (int64_t v0 = (int64_t)blockIdx.x * VL + vl; v0 < nvox; v0 += stride * U) {
        float4 xv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {   // out-of-range voxels re-read the last one (every lane takes part in the exchanges below)
            const int64_t v = v0 + u * stride;
            xv[u] = ldf4(x + (v < nvox ? v : nvox - 1) * x_ld + 4 * q);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t v = v0 + u * stride;
            float acc[CO];
#pragma unroll
            for (int co = 0; co < CO; ++co) {
                float a = xv[u].x * wq[co].x;
                a = fmaf(xv[u].y, wq[co].y, a);
                a = fmaf(xv[u].z, wq[co].z, a);
                a = fmaf(xv[u].w, wq[co].w, a);
                for (int d = 1; d < QC; d <<= 1) a += __shfl_xor(a, d, 64);   // the voxel's quads are QC neighbouring lanes
                acc[co] = a + bq[co];
            }
            if (q == 0 && v < nvox) {
                T* yp = y + v * y_ld;
#pragma unroll
                for (int co = 0; co < CO; ++co)
                    if (co < Co) stf(yp + co, acc[co]);
            }
        }
    }
}

constexpr int kPwMaxBlocks = 1024;

// part[blk][co][ci] (+ bias_part[blk][co])
// VX = input channels per lane: 4 (fp32: 16-byte loads) or 8 (bf16: 16-byte loads; with 4 the bf16 kernel moved 8 bytes per lane
// plus two 2-byte dy loads per voxel and ran at 1.5 TB/s, slower in absolute time than the fp32 one).  DYV: dy is dense with
// exactly CO channels, so a voxel's dy is ONE vector load.
template <typename T, int CO, int VX, bool DYV>
__global__ void __launch_bounds__(256)
pw_wgrad_kernel(const T* __restrict__ x, const T* __restrict__ dy, double* __restrict__ part,
                double* __restrict__ bias_part, int64_t nvox, int Ci, int Co, int x_ld, int y_ld) {
    extern __shared__ __attribute__((aligned(16))) double redd[];  // [256][VX*CO + CO]
    const int QC = Ci / VX;
    const int VL = 256 / QC;
    const int tid = threadIdx.x;
    const int q = tid % QC, vl = tid / QC;
    double acc[VX][CO], bsum[CO];
#pragma unroll
    for (int c = 0; c < CO; ++c) {
        bsum[c] = 0.0;
#pragma unroll
        for (int a = 0; a < VX; ++a) acc[a][c] = 0.0;
    }
    if (vl < VL) {
        constexpr int U = 4;
        const int64_t stride = (int64_t)gridDim.x * VL;
        for (int64_t v0 = (int64_t)blockIdx.x * VL + vl; v0 < nvox; v0 += stride * U) {
            float xv[U][VX];
            float gv[U][CO];
#pragma unroll
            for (int u = 0; u < U; ++u) {   // issue all loads of the 4 voxels first
                const int64_t v = v0 + u * stride;
                const bool ok = v < nvox;
                const T* xp = x + (ok ? v : 0) * x_ld + VX * q;
                if constexpr (VX == 8) {
                    const bf16x8_t t = *reinterpret_cast<const bf16x8_t*>(xp);
#pragma unroll
                    for (int a = 0; a < 8; ++a) xv[u][a] = ok ? (float)t[a] : 0.f;
                } else {
                    const float4 t = ldf4(xp);
                    xv[u][0] = ok ? t.x : 0.f, xv[u][1] = ok ? t.y : 0.f, xv[u][2] = ok ? t.z : 0.f, xv[u][3] = ok ? t.w : 0.f;
                }
                const T* gp = dy + (ok ? v : 0) * y_ld;
                if constexpr (DYV) {
                    typedef T tvec __attribute__((ext_vector_type(CO)));
                    const tvec t = *reinterpret_cast<const tvec*>(gp);
#pragma unroll
                    for (int c = 0; c < CO; ++c) gv[u][c] = ok ? (float)t[c] : 0.f;
                } else {
#pragma unroll
                    for (int c = 0; c < CO; ++c) gv[u][c] = (ok && c < Co) ? ldf(gp + c) : 0.f;
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
#pragma unroll
                for (int c = 0; c < CO; ++c) {
#pragma unroll
                    for (int a = 0; a < VX; ++a) acc[a][c] += (double)(xv[u][a] * gv[u][c]);
                    if (q == 0) bsum[c] += (double)gv[u][c];
                }
            }
        }
    }
    constexpr int S = (VX + 1) * CO;
#pragma unroll
    for (int c = 0; c < CO; ++c) {
#pragma unroll
        for (int a = 0; a < VX; ++a) redd[tid * S + a * CO + c] = acc[a][c];
        redd[tid * S + VX * CO + c] = bsum[c];
    }
    __syncthreads();
    // thread (q, a, c) sums over the VL voxel lanes
    for (int o = tid; o < QC * VX * CO; o += 256) {
        const int c = o % CO, a = (o / CO) % VX, qq = o / (VX * CO);
        double s = 0.0;
        for (int l = 0; l < VL; ++l) s += redd[(l * QC + qq) * S + a * CO + c];
        if (c < Co) part[((size_t)blockIdx.x * Co + c) * Ci + VX * qq + a] = s;
    }
    if (bias_part != nullptr && tid < Co) {
        double s = 0.0;
        for (int l = 0; l < VL; ++l) s += redd[(l * QC) * S + VX * CO + tid];
        bias_part[(size_t)blockIdx.x * Co + tid] = s;
    }
}

// one block per output element group: 256 threads = 8 elements x 32 partial lanes, fixed-order combine in double
__global__ void __launch_bounds__(256)
pw_wgrad_reduce_kernel(const double* __restrict__ part, const double* __restrict__ bias_part, float* __restrict__ dw,
                       float* __restrict__ dbias, int nb, int Ci, int Co) {
    __shared__ double red[256];
    const int el = threadIdx.x >> 5, ql = threadIdx.x & 31;
    const int nw = Co * Ci, ntot = nw + (dbias != nullptr ? Co : 0);
    const int i = blockIdx.x * 8 + el;
    double s = 0.0;
    if (i < nw)
        for (int b = ql; b < nb; b += 32) s += part[(size_t)b * nw + i];
    else if (i < ntot)
        for (int b = ql; b < nb; b += 32) s += bias_part[(size_t)b * Co + (i - nw)];
    red[threadIdx.x] = s;
    __syncthreads();
    if (ql == 0 && i < ntot) {
        double t = 0.0;
        for (int k = 0; k < 32; ++k) t += red[el * 32 + k];
        if (i < nw) dw[i] = (float)t;
        else dbias[i - nw] = (float)t;
    }
}

static bool is_pointwise(const Mri3dConvGeom& g) {
    return g.kd == 1 && g.kh == 1 && g.kw == 1 && g.sd == 1 && g.sh == 1 && g.sw == 1 && g.pd == 0 && g.ph == 0 &&
           g.pw == 0;
}

static int pw_blocks(const Mri3dConvGeom& g) {
    const int64_t nvox = (int64_t)g.n * g.dout * g.ho * g.wo;
    const int VL = 256 / (g.ci >> 2);
    int64_t want = cdiv64(nvox, (int64_t)VL * 16);
    return (int)std::max<int64_t>(1, std::min<int64_t>(want, kPwMaxBlocks));
}

bool conv_pointwise_supported(const Mri3dConvGeom& g, int pass) {
    if (!is_pointwise(g) || g.ci % 4 != 0 || g.x_ld % 4 != 0 || g.ci > 256) return false;
    if (pass == MRI3D_PASS_DGRAD) return g.co <= 8 && 256 % (g.ci >> 2) == 0;
    if (pass == MRI3D_PASS_FWD) {   // QC = Ci/4 lanes per voxel: a power of two inside one 64-lane wave
        const int qc = g.ci >> 2;
        return g.co <= 4 && qc >= 1 && qc <= 16 && (qc & (qc - 1)) == 0;
    }
    if (pass == MRI3D_PASS_WGRAD) return g.co <= 8 && g.ci <= 64;
    return false;
}

size_t conv_pointwise_workspace_bytes(const Mri3dConvGeom& g, int pass) {
    if (pass != MRI3D_PASS_WGRAD) return 0;
    return (size_t)pw_blocks(g) * (g.co * g.ci + g.co) * sizeof(double);
}

int conv_pointwise_dgrad(const Mri3dConvGeom& g, const void* dy, const float* w, const float* bias, void* dx,
                         hipStream_t s) {
    MRI3D_REQUIRE(aligned_vec4(g.dtype, dx), MRI3D_EINVAL, "conv3d_dgrad(pointwise): dx must be aligned to 4 elements");
    const int64_t nvox = (int64_t)g.n * g.di * g.hi * g.wi;
    const int VL = 256 / (g.ci >> 2);
    const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(cdiv64(nvox, (int64_t)VL * 4), 2048));
#define PW_DGRAD(CO)                                                                                                   \
    hipLaunchKernelGGL((pw_dgrad_kernel<T, CO>), dim3(grid), dim3(256), 0, s, (const T*)dy, w, bias, (T*)dx, nvox, g.ci, g.co, \
                       g.x_ld, g.y_ld)
    MRI3D_DISPATCH_DTYPE(g.dtype, T, {
        if (g.co <= 2) PW_DGRAD(2);
        else if (g.co <= 4) PW_DGRAD(4);
        else PW_DGRAD(8);
    });
#undef PW_DGRAD
    return check_launch("conv3d_dgrad(pointwise)");
}

int conv_pointwise_fwd(const Mri3dConvGeom& g, const void* x, const float* w, const float* bias, void* y, hipStream_t s) {
    MRI3D_REQUIRE(aligned_vec4(g.dtype, x), MRI3D_EINVAL, "conv3d_fwd(pointwise): x must be aligned to 4 elements");
    const int64_t nvox = (int64_t)g.n * g.di * g.hi * g.wi;
    const int VL = 256 / (g.ci >> 2);
    const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(cdiv64(nvox, (int64_t)VL * 4), 2048));
#define PW_FWD(CO)                                                                                                     \
    hipLaunchKernelGGL((pw_fwd_kernel<T, CO>), dim3(grid), dim3(256), 0, s, (const T*)x, w, bias, (T*)y, nvox, g.ci, g.co,     \
                       g.x_ld, g.y_ld)
    MRI3D_DISPATCH_DTYPE(g.dtype, T, {
        if (g.co <= 2) PW_FWD(2);
        else PW_FWD(4);
    });
#undef PW_FWD
    return check_launch("conv3d_fwd(pointwise)");
}

int conv_pointwise_wgrad(const Mri3dConvGeom& g, const void* x, const void* dy, float* dw, float* dbias, void* ws,
                         size_t ws_bytes, hipStream_t s) {
    const int nb = pw_blocks(g);
    const size_t need = conv_pointwise_workspace_bytes(g, MRI3D_PASS_WGRAD);
    MRI3D_REQUIRE(ws && ws_bytes >= need, MRI3D_EWORKSPACE, "conv3d_wgrad(pointwise): workspace %zu < %zu", ws_bytes, need);
    MRI3D_REQUIRE(aligned_vec4(g.dtype, x) && (reinterpret_cast<uintptr_t>(ws) & 7) == 0, MRI3D_EINVAL,
                  "conv3d_wgrad(pointwise): x must be aligned to 4 elements");
    double* part = static_cast<double*>(ws);
    double* bias_part = dbias ? part + (size_t)nb * g.co * g.ci : nullptr;
    const int64_t nvox = (int64_t)g.n * g.di * g.hi * g.wi;
    const int CO = g.co <= 2 ? 2 : (g.co <= 4 ? 4 : 8);
    // bf16: 8 channels per lane when the channel count and pitch allow 16-byte loads; one vector load of dy when it is dense
    const bool v8 = g.dtype == MRI3D_BF16 && g.ci % 8 == 0 && g.x_ld % 8 == 0 && 256 % (g.ci / 8) == 0 && aligned16(x) && CO <= 4;
    const bool dyv = g.co == CO && g.y_ld == CO && (reinterpret_cast<uintptr_t>(dy) & 15) == 0;
    const size_t smem = (size_t)256 * ((v8 ? 8 : 4) + 1) * CO * sizeof(double);
#define PW_WGRAD(CO_, VX_, DV_)                                                                                        \
    do {                                                                                                               \
        auto kern = pw_wgrad_kernel<T, CO_, VX_, DV_>;                                                                 \
        static const hipError_t attr_ = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                       \
                                                            hipFuncAttributeMaxDynamicSharedMemorySize, 256 * 9 * 8 * 8); \
        (void)attr_;                                                                                                   \
        hipLaunchKernelGGL(kern, dim3(nb), dim3(256), smem, s, (const T*)x, (const T*)dy, part, bias_part, nvox, g.ci,  \
                           g.co, g.x_ld, g.y_ld);                                                                      \
    } while (0)
#define PW_WGRAD_CO(CO_)                                                                                               \
    do {                                                                                                               \
        if constexpr (sizeof(T) == 2 && CO_ <= 4) {                                                                    \
            if (v8 && dyv) { PW_WGRAD(CO_, 8, true); break; }                                                          \
            if (v8) { PW_WGRAD(CO_, 8, false); break; }                                                                \
        }                                                                                                              \
        if (dyv) PW_WGRAD(CO_, 4, true);                                                                               \
        else PW_WGRAD(CO_, 4, false);                                                                                  \
    } while (0)
    MRI3D_DISPATCH_DTYPE(g.dtype, T, {
        if (CO == 2) PW_WGRAD_CO(2);
        else if (CO == 4) PW_WGRAD_CO(4);
        else PW_WGRAD_CO(8);
    });
#undef PW_WGRAD_CO
#undef PW_WGRAD
    hipLaunchKernelGGL(pw_wgrad_reduce_kernel, dim3(cdiv(g.co * g.ci + g.co, 8)), dim3(256), 0, s, part, bias_part, dw, dbias,
                       nb, g.ci, g.co);
    return check_launch("conv3d_wgrad(pointwise)");
}

}  // namespace mri3d
