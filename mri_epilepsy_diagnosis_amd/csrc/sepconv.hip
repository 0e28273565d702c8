// sepconv.hip — the head of the autoencoder's first DownBlock (/root/reference/classification/models/AE_model.py:45-53 with the
// shipped kwargs conv_k=6, conv_s=2, conv_pad=2): Conv3d(1, C, (k,1,1), stride (s,1,1)) followed by Conv3d(C, C2, (1,k,1), stride
// (1,s,1)) on the full-resolution input.  The intermediate a1 = convx(x) is the largest tensor of the model (C x D/2 x H x W: 315 MB
// for 4 volumes) and in the backward pass its gradient da1 = convy^T(da2) exists only to be contracted with the one-channel input:
//     dWx[c][kd] = sum_v x(s d1 + kd - p, h, w) da1[c](d1, h, w),   dbx[c] = sum_v da1[c]
// written by the data gradient of convy (157 MB read, 315 MB written) and read back by the weight gradient of convx (315 + 79 MB),
// 0.41 ms of the step.  mri3d_convpair_wgrad_first forms da1 per voxel in registers and contracts it on the spot: x and da2 are
// read once (through the caches), da1 never reaches memory.
#include "common.h"

namespace mri3d {

constexpr int kCpBlocks = 1024;

// One lane = one voxel of a1 at a time; a wave works on 64 consecutive w of one (n, d1, h) row, so the set of convy taps that
// reach the row (kh = (h + p2) mod s2, + s2, ...: at most MT of them) and every x plane index are wave-uniform and all index
// arithmetic is 32-bit and scalar.  Every load of a voxel — da2 at its taps, x at its K1 planes — is issued before the first
// multiply: with the loads inside the tap loop a voxel cost four dependent round trips (0.265 ms for 4 volumes; pairing rows to
// halve the weight reads made it 0.34 ms: 173 registers, two waves per SIMD).  Wy sits in LDS transposed to [kh][c2][c]
// (uniform-address reads).  C * K1 + C accumulators per lane; shuffle trees + the four waves in a fixed order at the end, one partial
// per workgroup, convpair_reduce_kernel (double) after it.
template <typename T, int C, int C2, int K1, int MT>
__global__ void __launch_bounds__(256)
convpair_wgrad_first_kernel(const T* __restrict__ x, const T* __restrict__ da2, const float* __restrict__ wy, float* __restrict__ part,
                            int N, int D, int H, int W, int D1, int H2, int s1, int p1, int K2, int s2, int p2, int x_ld, int d_ld) {
    constexpr int NA = C * K1 + C;
    __shared__ __attribute__((aligned(16))) float wl[8 * C2 * C];   // K2 <= 8
    __shared__ float wred[4][NA];
    for (int i = threadIdx.x; i < K2 * C2 * C; i += 256) {
        const int c = i % C, c2 = (i / C) % C2, kh = i / (C * C2);
        wl[i] = wy[((size_t)c2 * C + c) * K2 + kh];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float acc[C * K1], bacc[C];
#pragma unroll
    for (int i = 0; i < C * K1; ++i) acc[i] = 0.f;
#pragma unroll
    for (int i = 0; i < C; ++i) bacc[i] = 0.f;
    const int chunks = (W + 63) >> 6;
    const int rows = N * D1 * H;
    const int64_t xplane = (int64_t)H * W * x_ld;
    for (int row = blockIdx.x * 4 + wv; row < rows; row += gridDim.x * 4) {   // wave-uniform
        const int h = row % H, r2 = row / H;
        const int d1 = r2 % D1, n = r2 / D1;
        // the convy taps that read row h: kh with h + p2 - kh = s2 * h2, 0 <= h2 < H2   (uniform)
        int tkh[MT], th2[MT];
        unsigned tok = 0;
        {
            int kh = (h + p2) % s2;
#pragma unroll
            for (int t = 0; t < MT; ++t, kh += s2) {
                const int u = h + p2 - kh;
                const bool ok = kh < K2 && u >= 0 && u / s2 < H2;
                tok |= ok ? (1u << t) : 0u;
                tkh[t] = ok ? kh : 0, th2[t] = ok ? u / s2 : 0;
            }
        }
        const T* dn = da2 + ((int64_t)n * D1 + d1) * H2 * W * d_ld;
        const T* xr = x + (int64_t)n * D * xplane + (int64_t)h * W * x_ld;
        for (int ck = 0; ck < chunks; ++ck) {
            const int w = ck * 64 + lane;
            const bool live = w < W;
            const int wc = live ? w : 0;
            // ---- every load of the voxel first
            float4 q[MT][C2 / 4];
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int j = 0; j < C2 / 4; ++j) q[t][j] = ldf4(dn + ((int64_t)th2[t] * W + wc) * d_ld + 4 * j);   // (a skipped tap re-reads row 0)
            float xv[K1];
#pragma unroll
            for (int kd = 0; kd < K1; ++kd) {
                const int id = d1 * s1 + kd - p1;
                xv[kd] = ldf(xr + (int64_t)((unsigned)id < (unsigned)D ? id : 0) * xplane + (int64_t)wc * x_ld);
                if ((unsigned)id >= (unsigned)D) xv[kd] = 0.f;   // uniform
            }
            // ---- da1[c](d1, h, w)
            float g[C];
#pragma unroll
            for (int c = 0; c < C; ++c) g[c] = 0.f;
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                if ((tok >> t) & 1u) {   // uniform
                    const float* wk = wl + tkh[t] * C2 * C;
#pragma unroll
                    for (int j = 0; j < C2 / 4; ++j) {
                        const float qv[4] = {q[t][j].x, q[t][j].y, q[t][j].z, q[t][j].w};
#pragma unroll
                        for (int k = 0; k < 4; ++k)
#pragma unroll
                            for (int c = 0; c < C; c += 4) {
                                const float4 ww = *reinterpret_cast<const float4*>(wk + (4 * j + k) * C + c);
                                g[c] = fmaf(qv[k], ww.x, g[c]), g[c + 1] = fmaf(qv[k], ww.y, g[c + 1]);
                                g[c + 2] = fmaf(qv[k], ww.z, g[c + 2]), g[c + 3] = fmaf(qv[k], ww.w, g[c + 3]);
                            }
                    }
                }
            }
            if (!live) {
#pragma unroll
                for (int c = 0; c < C; ++c) g[c] = 0.f;
            }
#pragma unroll
            for (int c = 0; c < C; ++c) bacc[c] += g[c];
#pragma unroll
            for (int kd = 0; kd < K1; ++kd)
#pragma unroll
                for (int c = 0; c < C; ++c) acc[c * K1 + kd] = fmaf(xv[kd], g[c], acc[c * K1 + kd]);
        }
    }
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        float v = i < C * K1 ? acc[i < C * K1 ? i : 0] : bacc[i < C * K1 ? 0 : i - C * K1];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane == 0) wred[wv][i] = v;
    }
    __syncthreads();
    float* out = part + (size_t)blockIdx.x * NA;
    for (int i = threadIdx.x; i < NA; i += 256) out[i] = (wred[0][i] + wred[1][i]) + (wred[2][i] + wred[3][i]);
}

// out[i] = sum_b part[b][i]  (double, fixed order); the first nw elements are dW, the rest dbias (may be NULL)
__global__ void __launch_bounds__(64)
convpair_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, float* __restrict__ dbias, int nb, int nw, int nbias) {
    const int i = blockIdx.x;
    double s = 0.0;
    for (int b = threadIdx.x; b < nb; b += 64) s += (double)part[(size_t)b * (nw + nbias) + i];
    __shared__ double red[64];
    red[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int k = 0; k < 64; ++k) t += red[k];
        if (i < nw) dw[i] = (float)t;
        else if (dbias != nullptr) dbias[i - nw] = (float)t;
    }
}

// first: Conv3d(1, C, (K1,1,1), stride (s1,1,1), pad (p1,0,0)); second: Conv3d(C, C2, (1,K2,1), stride (1,s2,1), pad (0,p2,0)) on its output
static bool convpair_ok(const Mri3dConvGeom& a, const Mri3dConvGeom& b) {
    const bool first = a.ci == 1 && a.kh == 1 && a.kw == 1 && a.sh == 1 && a.sw == 1 && a.ph == 0 && a.pw == 0 && a.dd == 1 && a.dh == 1 &&
                       a.dw == 1 && a.x_ld >= 1 && a.sd >= 1 && a.kd == 6;
    const bool second = b.kd == 1 && b.kw == 1 && b.sd == 1 && b.sw == 1 && b.pd == 0 && b.pw == 0 && b.dd == 1 && b.dh == 1 && b.dw == 1 &&
                        b.kh >= 1 && b.kh <= 8 && b.sh >= 1 && cdiv(b.kh, b.sh) <= 3 && b.ci == a.co && b.n == a.n && b.di == a.dout && b.hi == a.ho && b.wi == a.wo;
    return first && second && a.co == 8 && b.co == 8 && b.y_ld % 4 == 0 && b.y_ld >= b.co && a.dtype == b.dtype &&
           (a.dtype == MRI3D_F32 || a.dtype == MRI3D_BF16) && a.n > 0;
}

}  // namespace mri3d

using namespace mri3d;

extern "C" int32_t mri3d_convpair_supported(const Mri3dConvGeom* first, const Mri3dConvGeom* second) {
    return first && second && convpair_ok(*first, *second) ? 1 : 0;
}

extern "C" size_t mri3d_convpair_workspace_bytes(const Mri3dConvGeom* first, const Mri3dConvGeom* second) {
    if (!first || !second || !convpair_ok(*first, *second)) return 0;
    return (size_t)kCpBlocks * (first->co * first->kd + first->co) * sizeof(float);
}

extern "C" int mri3d_convpair_wgrad_first(const Mri3dConvGeom* first, const Mri3dConvGeom* second, const void* x, const void* dy2,
                                          const float* w2, float* dw1, float* dbias1, void* workspace, size_t ws_bytes,
                                          mri3d_stream_t stream) {
    MRI3D_REQUIRE(first && second && x && dy2 && w2 && dw1, MRI3D_EINVAL, "convpair_wgrad_first: null pointer");
    MRI3D_REQUIRE(convpair_ok(*first, *second), MRI3D_ENOTSUP, "convpair_wgrad_first: geometry not served (see mri3d_convpair_supported)");
    MRI3D_REQUIRE(workspace && ws_bytes >= mri3d_convpair_workspace_bytes(first, second), MRI3D_EINVAL,
                  "convpair_wgrad_first: workspace too small");
    MRI3D_REQUIRE(aligned_vec4(first->dtype, dy2), MRI3D_EINVAL, "convpair_wgrad_first: dy2 must be aligned to four channels");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const Mri3dConvGeom &a = *first, &b = *second;
    const int64_t rows = (int64_t)a.n * a.dout * a.ho;   // (n, d1, h)
    MRI3D_REQUIRE(rows < 0x7ffffff0, MRI3D_ENOTSUP, "convpair_wgrad_first: too many rows");
    const int nb = (int)std::min<int64_t>(cdiv64(rows, 4), kCpBlocks);
    float* part = static_cast<float*>(workspace);
    MRI3D_DISPATCH_DTYPE(a.dtype, T, {
        hipLaunchKernelGGL((convpair_wgrad_first_kernel<T, 8, 8, 6, 3>), dim3(nb), dim3(256), 0, s, (const T*)x, (const T*)dy2, w2, part, a.n,
                           a.di, a.hi, a.wi, a.dout, b.ho, a.sd, a.pd, b.kh, b.sh, b.ph, a.x_ld, b.y_ld);
    });
    int rc = check_launch("convpair_wgrad_first");
    if (rc) return rc;
    const int nw = a.co * a.kd;
    hipLaunchKernelGGL(convpair_reduce_kernel, dim3(nw + a.co), dim3(64), 0, s, part, dw1, dbias1, nb, nw, a.co);
    return check_launch("convpair_wgrad_first(reduce)");
}
