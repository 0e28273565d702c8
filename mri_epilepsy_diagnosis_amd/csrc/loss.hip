// loss.hip — fused softmax(dim=C) + soft-Dice loss and its gradient, plus the argmax->uint8 mask used by validation.
// Reference: F.softmax(logits, dim=1) -> get_dice_loss(probabilities, targets) -> .mean()
// (segmentation/routine.py:239-253, 272-274); labels = logits.argmax(dim=1) (routine.py:226-227).
//
//   p = softmax(z);  tp_c = sum p_c g_c,  fp_c = sum p_c (1-g_c),  fn_c = sum (1-p_c) g_c
//   dice_c = 2 tp_c / (2 tp_c + fp_c + fn_c + eps) = 2 tp_c / (sum p_c + sum g_c + eps)
//   loss = mean_{n,c} (1 - dice_{n,c})
// The reference feeds a (N,1,...) target against (N,2,...) probabilities, i.e. g_c = g for every class (SURVEY
// Appendix C.1); ct == 1 reproduces that broadcast, ct == c is the per-class form.
//
// HBM-bound: forward = one read of logits+target; backward = one read of both + one write of dlogits.
#include "common.h"
#include <algorithm>

namespace mri3d {

constexpr int kDiceMaxC = 8;
constexpr int kDiceMaxBlocks = 1024;

template <int C>
__device__ __forceinline__ void softmax_c(const float* z, float (&p)[C]) {
    float m = z[0];
#pragma unroll
    for (int j = 1; j < C; ++j) m = fmaxf(m, z[j]);
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < C; ++j) { p[j] = expf(z[j] - m); s += p[j]; }
    float inv = 1.f / s;
#pragma unroll
    for (int j = 0; j < C; ++j) p[j] *= inv;
}

// part[n][blk][c][3] = (tp, sum_p, sum_g)
template <typename T, int C>
__global__ void __launch_bounds__(256)
dice_fwd_kernel(const T* __restrict__ logits, const float* __restrict__ target, double* __restrict__ part,
                int64_t vox, int ct, int x_ld, int t_ld) {
    __shared__ double red[4][C * 3];
    const int n = blockIdx.y;
    const T* zn = logits + (int64_t)n * vox * x_ld;
    const float* tn = target + (int64_t)n * vox * t_ld;
    double tp[C], sp[C], sg[C];  // torch's CPU reductions (the oracle's arithmetic) accumulate float sums in double
#pragma unroll
    for (int j = 0; j < C; ++j) { tp[j] = 0.0; sp[j] = 0.0; sg[j] = 0.0; }
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < vox; v += (int64_t)gridDim.x * blockDim.x) {
        float z[C], p[C];
#pragma unroll
        for (int j = 0; j < C; ++j) z[j] = ldf(zn + v * x_ld + j);
        softmax_c<C>(z, p);
#pragma unroll
        for (int j = 0; j < C; ++j) {
            float g = tn[v * t_ld + (ct == 1 ? 0 : j)];
            tp[j] += (double)(p[j] * g);
            sp[j] += (double)p[j];
            sg[j] += (double)g;
        }
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int j = 0; j < C; ++j) {
        double a = wave_sum_d(tp[j]), b = wave_sum_d(sp[j]), c = wave_sum_d(sg[j]);
        if (lane == 0) { red[wave][j * 3] = a; red[wave][j * 3 + 1] = b; red[wave][j * 3 + 2] = c; }
    }
    __syncthreads();
    if (threadIdx.x < C * 3) {
        double s = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        part[((size_t)n * gridDim.x + blockIdx.x) * C * 3 + threadIdx.x] = s;
    }
}

// one block of 256 threads: (n, c) pairs are handled 8 at a time by 32 partial lanes each (fixed-order double sums)
__global__ void __launch_bounds__(256)
dice_finalize_kernel(const double* __restrict__ part, float* __restrict__ stats, float* __restrict__ loss, int N, int C,
                     int nblk, float eps) {
    __shared__ double r0[256], r1[256], r2[256];
    __shared__ double acc_loss;
    const int el = threadIdx.x >> 5, ql = threadIdx.x & 31;
    if (threadIdx.x == 0) acc_loss = 0.0;
    __syncthreads();
    for (int base = 0; base < N * C; base += 8) {
        const int i = base + el;
        double tp = 0.0, sp = 0.0, sg = 0.0;
        if (i < N * C) {
            const int n = i / C, c = i - n * C;
            const double* p = part + ((size_t)n * nblk * C + c) * 3;
            for (int q = ql; q < nblk; q += 32) {
                tp += p[(size_t)q * C * 3];
                sp += p[(size_t)q * C * 3 + 1];
                sg += p[(size_t)q * C * 3 + 2];
            }
        }
        r0[threadIdx.x] = tp; r1[threadIdx.x] = sp; r2[threadIdx.x] = sg;
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int e = 0; e < 8 && base + e < N * C; ++e) {
                double a = 0.0, b = 0.0, d = 0.0;
                for (int k = 0; k < 32; ++k) { a += r0[e * 32 + k]; b += r1[e * 32 + k]; d += r2[e * 32 + k]; }
                const int i2 = base + e;
                stats[i2 * 3] = (float)a;
                stats[i2 * 3 + 1] = (float)b;
                stats[i2 * 3 + 2] = (float)d;
                // float arithmetic from here mirrors the reference's fp32 tensors
                const float ftp = (float)a, den = 2.f * ftp + ((float)b - ftp) + ((float)d - ftp) + eps;
                acc_loss += 1.0 - (double)(2.f * ftp / den);
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = (float)(acc_loss / (double)(N * C));
}

template <typename T, int C>
__global__ void __launch_bounds__(256)
dice_bwd_kernel(const T* __restrict__ logits, const float* __restrict__ target, const float* __restrict__ stats,
                const float* __restrict__ dloss, T* __restrict__ dlogits, int64_t vox, int N, int ct, int x_ld,
                int t_ld, float eps) {
    const int n = blockIdx.y;
    const T* zn = logits + (int64_t)n * vox * x_ld;
    const float* tn = target + (int64_t)n * vox * t_ld;
    T* dn = dlogits + (int64_t)n * vox * x_ld;
    // d(1 - dice)/dp_c(v) = -(2 g den - 2 tp)/den^2, scaled by dloss/(N*C)
    float ka[C], kb[C];
    const float scale = dloss[0] / (float)(N * C);
#pragma unroll
    for (int j = 0; j < C; ++j) {
        float tp = stats[(n * C + j) * 3], sp = stats[(n * C + j) * 3 + 1], sg = stats[(n * C + j) * 3 + 2];
        float den = sp + sg + eps;
        ka[j] = -2.f * scale / den;          // multiplies g
        kb[j] = 2.f * scale * tp / (den * den);  // constant term
    }
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < vox; v += (int64_t)gridDim.x * blockDim.x) {
        float z[C], p[C], dp[C];
#pragma unroll
        for (int j = 0; j < C; ++j) z[j] = ldf(zn + v * x_ld + j);
        softmax_c<C>(z, p);
        float dot = 0.f;
#pragma unroll
        for (int j = 0; j < C; ++j) {
            float g = tn[v * t_ld + (ct == 1 ? 0 : j)];
            dp[j] = fmaf(ka[j], g, kb[j]);
            dot = fmaf(p[j], dp[j], dot);
        }
#pragma unroll
        for (int j = 0; j < C; ++j) stf(dn + v * x_ld + j, p[j] * (dp[j] - dot));
    }
}

template <typename T>
__global__ void __launch_bounds__(256)
argmax_u8_kernel(const T* __restrict__ logits, uint8_t* __restrict__ out, int64_t nvox, int C, int ld) {
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvox; v += (int64_t)gridDim.x * blockDim.x) {
        const T* z = logits + v * ld;
        float best = ldf(z);
        int bi = 0;
        for (int j = 1; j < C; ++j) {
            float t = ldf(z + j);
            // torch.argmax: first maximal value; NaN counts as maximal
            if ((t > best && best == best) || (t != t && best == best)) { best = t; bi = j; }
        }
        out[v] = (uint8_t)bi;
    }
}

// ------------------------------------------------------------------ mask overlap counts (validation metrics)
// validate_dsc_asd scores the arg-max mask against the label map on the host (segmentation/routine.py:198-235,
// metrics.py:312-329): Dice = 2*sum(gt & pred) / (sum(gt) + sum(pred)), IoU = #(pred>0 and gt>0) / #(pred>0 or gt>0).
// All five sums are integers, so they are taken on the device in one pass over the two uint8 masks (16 voxels per lane
// per load) and only 40 bytes leave the GPU instead of two volumes; integer addition is order-independent => exact.
//   out[0] = sum(gt)  out[1] = sum(pred)  out[2] = sum(gt & pred)  out[3] = #(gt>0 & pred>0)  out[4] = #(gt>0 | pred>0)
constexpr int kOvlBlocks = 512;

__device__ __forceinline__ void ovl_acc(unsigned g, unsigned p, unsigned long long (&c)[5]) {
    c[0] += g;
    c[1] += p;
    c[2] += g & p;
    c[3] += (g != 0u && p != 0u) ? 1u : 0u;
    c[4] += (g != 0u || p != 0u) ? 1u : 0u;
}

__global__ void __launch_bounds__(256)
mask_overlap_kernel(const uint8_t* __restrict__ pred, const uint8_t* __restrict__ gt, int64_t n, int vec,
                    unsigned long long* __restrict__ part) {
    __shared__ unsigned long long red[4][5];
    unsigned long long c[5] = {0ull, 0ull, 0ull, 0ull, 0ull};
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (int64_t)gridDim.x * blockDim.x;
    int64_t done = 0;
    if (vec) {
        const int64_t n16 = n >> 4;
        const uint4* p4 = reinterpret_cast<const uint4*>(pred);
        const uint4* g4 = reinterpret_cast<const uint4*>(gt);
        for (int64_t i = tid; i < n16; i += nth) {
            const uint4 pv = p4[i], gv = g4[i];
            const unsigned pw[4] = {pv.x, pv.y, pv.z, pv.w}, gw[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
            for (int w = 0; w < 4; ++w)
#pragma unroll
                for (int b = 0; b < 4; ++b) ovl_acc((gw[w] >> (8 * b)) & 0xffu, (pw[w] >> (8 * b)) & 0xffu, c);
        }
        done = n16 << 4;
    }
    for (int64_t i = done + tid; i < n; i += nth) ovl_acc(gt[i], pred[i], c);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        unsigned long long v = c[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        if (lane == 0) red[wave][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < 5)
        part[(size_t)blockIdx.x * 5 + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

__global__ void mask_overlap_finalize_kernel(const unsigned long long* __restrict__ part, int nblk, int64_t* __restrict__ out) {
    const int k = threadIdx.x;
    if (k >= 5) return;
    unsigned long long s = 0ull;
    for (int b = 0; b < nblk; ++b) s += part[(size_t)b * 5 + k];
    out[k] = (int64_t)s;
}

static int dice_blocks(const Mri3dDiceGeom& g) {
    int64_t want = cdiv64(g.vox, 256 * 4);
    int cap = kDiceMaxBlocks / g.n;
    if (cap < 1) cap = 1;
    int b = (int)(want < cap ? want : cap);
    return b < 1 ? 1 : b;
}

}  // namespace mri3d

using namespace mri3d;

static int dice_check(const Mri3dDiceGeom* g, const char* who) {
    MRI3D_REQUIRE(g != nullptr, MRI3D_EINVAL, "%s: null geometry", who);
    MRI3D_REQUIRE(g->dtype == MRI3D_F32 || g->dtype == MRI3D_BF16, MRI3D_ENOTSUP, "%s: unknown dtype %d", who, g->dtype);
    MRI3D_REQUIRE(g->n > 0 && g->vox > 0, MRI3D_EINVAL, "%s: empty tensor", who);
    MRI3D_REQUIRE(g->c >= 2 && g->c <= kDiceMaxC, MRI3D_ENOTSUP, "%s: classes must be in [2,%d], got %d", who, kDiceMaxC,
                  g->c);
    MRI3D_REQUIRE(g->ct == 1 || g->ct == g->c, MRI3D_EINVAL, "%s: target channels %d must be 1 or %d", who, g->ct, g->c);
    MRI3D_REQUIRE(g->x_ld >= g->c && g->t_ld >= g->ct, MRI3D_EINVAL, "%s: bad pitch", who);
    return MRI3D_OK;
}

extern "C" size_t mri3d_softmax_dice_workspace_bytes(const Mri3dDiceGeom* g) {
    if (!g) return 0;
    return (size_t)(kDiceMaxBlocks + g->n) * g->c * 3 * sizeof(double);
}

#define DICE_DISPATCH(CALL)                 \
    switch (g->c) {                         \
        case 2: CALL(2); break;             \
        case 3: CALL(3); break;             \
        case 4: CALL(4); break;             \
        case 5: CALL(5); break;             \
        case 6: CALL(6); break;             \
        case 7: CALL(7); break;             \
        default: CALL(8); break;            \
    }

extern "C" int mri3d_softmax_dice_fwd(const Mri3dDiceGeom* g, const void* logits, const void* target, float* loss,
                                      float* stats, void* workspace, size_t ws_bytes, mri3d_stream_t stream) {
    int rc = dice_check(g, "softmax_dice_fwd");
    if (rc) return rc;
    MRI3D_REQUIRE(logits && target && loss && stats, MRI3D_EINVAL, "softmax_dice_fwd: null pointer");
    MRI3D_REQUIRE(workspace && ws_bytes >= mri3d_softmax_dice_workspace_bytes(g), MRI3D_EWORKSPACE,
                  "softmax_dice_fwd: workspace %zu < %zu", ws_bytes, mri3d_softmax_dice_workspace_bytes(g));
    hipStream_t s = static_cast<hipStream_t>(stream);
    int nblk = dice_blocks(*g);
    double* part = static_cast<double*>(workspace);
#define CALL(CC)                                                                                              \
    hipLaunchKernelGGL((dice_fwd_kernel<T, CC>), dim3(nblk, g->n), dim3(256), 0, s, (const T*)logits,         \
                       (const float*)target, part, g->vox, g->ct, g->x_ld, g->t_ld)
    MRI3D_DISPATCH_DTYPE(g->dtype, T, { DICE_DISPATCH(CALL) });
#undef CALL
    hipLaunchKernelGGL(dice_finalize_kernel, dim3(1), dim3(256), 0, s, part, stats, loss, g->n, g->c, nblk, g->eps);
    return check_launch("softmax_dice_fwd");
}

extern "C" int mri3d_softmax_dice_bwd(const Mri3dDiceGeom* g, const void* logits, const void* target,
                                      const float* stats, const float* dloss, void* dlogits, mri3d_stream_t stream) {
    int rc = dice_check(g, "softmax_dice_bwd");
    if (rc) return rc;
    MRI3D_REQUIRE(logits && target && stats && dloss && dlogits, MRI3D_EINVAL, "softmax_dice_bwd: null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    int nblk = stream_grid(g->vox, 256);
    if (nblk * g->n > 4096) nblk = 4096 / g->n > 0 ? 4096 / g->n : 1;
#define CALL(CC)                                                                                              \
    hipLaunchKernelGGL((dice_bwd_kernel<T, CC>), dim3(nblk, g->n), dim3(256), 0, s, (const T*)logits,         \
                       (const float*)target, stats, dloss, (T*)dlogits, g->vox, g->n, g->ct, g->x_ld,        \
                       g->t_ld, g->eps)
    MRI3D_DISPATCH_DTYPE(g->dtype, T, { DICE_DISPATCH(CALL) });
#undef CALL
    return check_launch("softmax_dice_bwd");
}

extern "C" int mri3d_argmax_u8(const void* logits, uint8_t* out, int64_t nvox, int32_t c, int32_t ld, int32_t dtype,
                               mri3d_stream_t stream) {
    MRI3D_REQUIRE(dtype == MRI3D_F32 || dtype == MRI3D_BF16, MRI3D_ENOTSUP, "argmax_u8: unknown dtype %d", dtype);
    MRI3D_REQUIRE(logits && out && nvox > 0 && c > 0 && c <= 256 && ld >= c, MRI3D_EINVAL, "argmax_u8: bad arguments");
    hipStream_t s = static_cast<hipStream_t>(stream);
    MRI3D_DISPATCH_DTYPE(dtype, T, {
        hipLaunchKernelGGL(argmax_u8_kernel<T>, dim3(stream_grid(nvox, 256)), dim3(256), 0, s, (const T*)logits, out, nvox, c,
                           ld);
    });
    return check_launch("argmax_u8");
}

extern "C" size_t mri3d_mask_overlap_workspace_bytes(void) { return (size_t)kOvlBlocks * 5 * sizeof(unsigned long long); }

extern "C" int mri3d_mask_overlap(const uint8_t* pred, const uint8_t* gt, int64_t nvox, int64_t* counts, void* workspace,
                                  size_t ws_bytes, mri3d_stream_t stream) {
    MRI3D_REQUIRE(pred && gt && counts && nvox > 0, MRI3D_EINVAL, "mask_overlap: bad arguments");
    MRI3D_REQUIRE(workspace && ws_bytes >= mri3d_mask_overlap_workspace_bytes() &&
                      (reinterpret_cast<uintptr_t>(workspace) & 7) == 0,
                  MRI3D_EWORKSPACE, "mask_overlap: workspace %zu < %zu", ws_bytes, mri3d_mask_overlap_workspace_bytes());
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int vec = ((reinterpret_cast<uintptr_t>(pred) | reinterpret_cast<uintptr_t>(gt)) & 15) == 0 ? 1 : 0;
    int nblk = (int)std::min<int64_t>(kOvlBlocks, cdiv64(nvox, 256 * 16));
    if (nblk < 1) nblk = 1;
    unsigned long long* part = static_cast<unsigned long long*>(workspace);
    hipLaunchKernelGGL(mask_overlap_kernel, dim3(nblk), dim3(256), 0, s, pred, gt, nvox, vec, part);
    hipLaunchKernelGGL(mask_overlap_finalize_kernel, dim3(1), dim3(64), 0, s, part, nblk, counts);
    return check_launch("mask_overlap");
}
