// upconv.hip — Conv3d over a nearest-neighbour upsampled input that is never formed:
//     y = conv3d(upsample_nearest(x, scale_factor = S), w, b, stride 1, padding p)
// The last UpBlock of the autoencoder (/root/reference/classification/models/AE_model.py:110-120 with the shipped kwargs
// up='upsample', scale=4, scale_mode='nearest': nn.Upsample(4) -> Conv3d(8, 1, (3,1,1)) at 160x192x160) writes an 8-channel
// full-resolution tensor (629 MB for 4 volumes) that exists only to be read back by the next kernel: forward writes and reads
// it, backward writes its gradient (conv data gradient) and reads it twice (upsample backward, conv weight gradient) — 3.1 GB,
// 1.0 ms of the 4.6 ms step.  Here the convolution indexes the COARSE tensor (fine input voxel u -> coarse voxel u / S per axis;
// zero padding applies at the fine level, as in the reference), the data gradient is summed over each coarse voxel's S^3 fine
// voxels inside the kernel (a fixed shuffle tree: deterministic), and the weight gradient reads x through the same index map.
// All three passes then stream y / dy once (79 MB) and read the coarse tensor (10 MB) from cache.
//
// Geometry convention: the Mri3dConvGeom describes the convolution on the VIRTUAL fine input (di, hi, wi = S x the coarse
// extents, x_ld = the coarse tensor's pitch); stride 1, dilation 1, at most 8 taps, Ci % 4 == 0, taps * Ci * Co <= 64.
#include "common.h"

namespace mri3d {

constexpr int kUcMaxW = 64;      // taps * Ci * Co accumulators a lane holds in the weight gradient at most
constexpr int kUcBlocks = 1024;  // weight-gradient partials

struct UpConvTaps {
    int n, od[8], oh[8], ow[8];  // tap offsets (kd - pd, kh - ph, kw - pw)
};

__host__ __device__ inline UpConvTaps upconv_taps(const Mri3dConvGeom& g) {
    UpConvTaps t;
    t.n = g.kd * g.kh * g.kw;
    for (int i = 0; i < 8; ++i) {
        const int k = i < t.n ? i : 0;
        t.ow[i] = k % g.kw - g.pw, t.oh[i] = (k / g.kw) % g.kh - g.ph, t.od[i] = k / (g.kw * g.kh) - g.pd;
    }
    return t;
}

// ------------------------------------------------------------------ forward
// One lane = one fine output voxel, all Co output channels.  (Ci, Co, taps) and the scale are template parameters: the loops are
// unrolled, the index map is a shift, and the weights are wave-uniform loads the compiler keeps on the scalar path; coarse offsets
// are 32-bit (the coarse tensor is small).
template <typename T, int CI, int CO, int NT, int LS>   // LS = log2(scale)
__global__ void __launch_bounds__(256)
upconv_fwd_kernel(Mri3dConvGeom g, const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                  T* __restrict__ y, int hch) {
    const UpConvTaps tp = upconv_taps(g);
    const int dc = g.di >> LS, hc = g.hi >> LS, wc = g.wi >> LS;
    const int hchunks = (g.ho + hch - 1) / hch;
    const int slabs = g.n * g.dout * hchunks;
    for (int slab = blockIdx.x; slab < slabs; slab += gridDim.x) {
        const int hk = slab % hchunks, nd = slab / hchunks;
        const int n = nd / g.dout, od = nd - n * g.dout;
        const int h0 = hk * hch, hn = min(hch, g.ho - h0);
        const unsigned inner = (unsigned)hn * g.wo;
        const T* xn = x + (int64_t)n * dc * hc * wc * g.x_ld;
        T* yn = y + (((int64_t)nd * g.ho + h0) * g.wo) * g.y_ld;
        for (unsigned e = threadIdx.x; e < inner; e += 256) {
            const int ow = e % (unsigned)g.wo, oh = h0 + e / (unsigned)g.wo;
            float acc[CO];
#pragma unroll
            for (int j = 0; j < CO; ++j) acc[j] = bias != nullptr ? bias[j] : 0.f;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int id = od + tp.od[t], ih = oh + tp.oh[t], iw = ow + tp.ow[t];
                const bool ok = (unsigned)id < (unsigned)g.di && (unsigned)ih < (unsigned)g.hi && (unsigned)iw < (unsigned)g.wi;
                const int off = ok ? (((id >> LS) * hc + (ih >> LS)) * wc + (iw >> LS)) * g.x_ld : 0;
#pragma unroll
                for (int ci = 0; ci < CI; ci += 4) {
                    float4 q = ldf4(xn + off + ci);
                    if (!ok) q = make_float4(0.f, 0.f, 0.f, 0.f);
                    const float qv[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                    for (int k = 0; k < 4; ++k)
#pragma unroll
                        for (int j = 0; j < CO; ++j) acc[j] = fmaf(qv[k], w[(j * CI + ci + k) * NT + t], acc[j]);
                }
            }
            T* yp = yn + (int64_t)e * g.y_ld;
#pragma unroll
            for (int j = 0; j < CO; ++j) stf(yp + j, acc[j]);
        }
    }
}

// ------------------------------------------------------------------ data gradient, summed onto the coarse tensor
// The S^3 fine voxels of a coarse voxel sit on consecutive lanes (scale 4: one coarse voxel per wave; scale 2: eight).  A lane forms
// the data gradient of its fine voxel for all Ci input channels, and the lanes of a coarse voxel are summed by a butterfly in a
// fixed order: while a lane holds more than one channel a step halves its channels (it keeps one half and hands the other to its
// partner: reduce-scatter), the remaining steps add single values.  A workgroup walks coarse rows (n, cd, ch): every coordinate but
// the lane's own sub-voxel is scalar.
template <int N>
__device__ __forceinline__ void halve_channels(float (&v)[N], int bit, int lane, int& c0) {
    if constexpr (N >= 2) {
        const bool hi = lane & bit;
#pragma unroll
        for (int k = 0; k < N / 2; ++k) {
            const float send = hi ? v[k] : v[k + N / 2], keep = hi ? v[k + N / 2] : v[k];
            v[k] = keep + __shfl_xor(send, bit, 64);
        }
        c0 += hi ? N / 2 : 0;
    }
}

template <typename T, int CI, int CO, int NT, int LS>
__global__ void __launch_bounds__(256)
upconv_dgrad_kernel(Mri3dConvGeom g, const T* __restrict__ dy, const float* __restrict__ w, T* __restrict__ dx, int rows) {
    constexpr int S = 1 << LS, SV = S * S * S, PER = 256 / SV;   // coarse voxels per workgroup pass
    const UpConvTaps tp = upconv_taps(g);
    const int dc = g.di >> LS, hc = g.hi >> LS, wc = g.wi >> LS;
    const int lane = threadIdx.x & 63;
    const int sub = threadIdx.x % SV, slot = threadIdx.x / SV;
    const int sw = sub & (S - 1), sh = (sub >> LS) & (S - 1), sd = sub >> (2 * LS);
    for (int row = blockIdx.x; row < rows; row += gridDim.x) {
        const int ch = row % hc, r2 = row / hc;
        const int cd = r2 % dc, n = r2 / dc;
        const int id = (cd << LS) + sd, ih = (ch << LS) + sh;   // the lane's fine INPUT voxel: (id, ih, iw)
        const T* dyn = dy + (int64_t)n * g.dout * g.ho * g.wo * g.y_ld;
        T* dxr = dx + ((int64_t)(n * dc + cd) * hc + ch) * wc * g.x_ld;
        int rowoff[NT];   // per tap: the output plane / row this lane reads is fixed for the coarse row
        unsigned rowok = 0;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int od = id - tp.od[t], oh = ih - tp.oh[t];   // y[o] reads x[o + off_t]: input voxel u feeds output o = u - off_t
            const bool ok = (unsigned)od < (unsigned)g.dout && (unsigned)oh < (unsigned)g.ho;
            rowok |= ok ? (1u << t) : 0u;
            rowoff[t] = ok ? (od * g.ho + oh) * g.wo : 0;
        }
        for (int cw0 = 0; cw0 < wc; cw0 += PER) {
            const int cw = cw0 + slot;
            const bool live = cw < wc;
            const int iw = (cw << LS) + sw;
            float a[CI];
#pragma unroll
            for (int k = 0; k < CI; ++k) a[k] = 0.f;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int ow = iw - tp.ow[t];
                const bool ok = live && ((rowok >> t) & 1u) && (unsigned)ow < (unsigned)g.wo;
                const T* dp = dyn + (int64_t)(ok ? rowoff[t] + ow : 0) * g.y_ld;
#pragma unroll
                for (int co = 0; co < CO; ++co) {
                    const float d = ok ? ldf(dp + co) : 0.f;
#pragma unroll
                    for (int k = 0; k < CI; ++k) a[k] = fmaf(d, w[(co * CI + k) * NT + t], a[k]);
                }
            }
            // butterfly over the SV lanes of the coarse voxel, high lane bit first
            int c0 = 0;
            constexpr int B0 = SV / 2;
            halve_channels<CI>(a, B0, lane, c0);
            float (&a1)[CI >= 2 ? CI / 2 : 1] = reinterpret_cast<float (&)[CI >= 2 ? CI / 2 : 1]>(a);
            constexpr int N1 = CI >= 2 ? CI / 2 : 1, B1 = B0 / 2;
            if constexpr (B1 >= 1) halve_channels<N1>(a1, B1, lane, c0);
            float (&a2)[N1 >= 2 ? N1 / 2 : 1] = reinterpret_cast<float (&)[N1 >= 2 ? N1 / 2 : 1]>(a);
            constexpr int N2 = N1 >= 2 ? N1 / 2 : 1, B2 = B1 / 2;
            if constexpr (B2 >= 1) halve_channels<N2>(a2, B2, lane, c0);
            float (&a3)[N2 >= 2 ? N2 / 2 : 1] = reinterpret_cast<float (&)[N2 >= 2 ? N2 / 2 : 1]>(a);
            constexpr int N3 = N2 >= 2 ? N2 / 2 : 1, B3 = B2 / 2;
            if constexpr (B3 >= 1) halve_channels<N3>(a3, B3, lane, c0);
            float (&a4)[N3 >= 2 ? N3 / 2 : 1] = reinterpret_cast<float (&)[N3 >= 2 ? N3 / 2 : 1]>(a);
            constexpr int N4 = N3 >= 2 ? N3 / 2 : 1, B4 = B3 / 2;
            if constexpr (B4 >= 1) halve_channels<N4>(a4, B4, lane, c0);
            float (&a5)[N4 >= 2 ? N4 / 2 : 1] = reinterpret_cast<float (&)[N4 >= 2 ? N4 / 2 : 1]>(a);
            constexpr int N5 = N4 >= 2 ? N4 / 2 : 1, B5 = B4 / 2;
            if constexpr (B5 >= 1) halve_channels<N5>(a5, B5, lane, c0);
            constexpr int N6 = N5 >= 2 ? N5 / 2 : 1;
            // steps that found a single channel per lane did not run (N = 1): sum those lane bits now.  A step ran iff its N >= 2.
            constexpr int ran = (CI >= 2) + (N1 >= 2 && B1 >= 1) + (N2 >= 2 && B2 >= 1) + (N3 >= 2 && B3 >= 1) + (N4 >= 2 && B4 >= 1) + (N5 >= 2 && B5 >= 1);
            constexpr int REM = SV >> ran;   // lanes still holding partial sums of the same channels: bits REM/2 .. 1
            constexpr int NF = CI >> ran;    // channels per lane at the end
            static_assert(NF >= 1 && N6 >= 1, "butterfly bookkeeping");
#pragma unroll
            for (int o = REM / 2; o >= 1; o >>= 1)
#pragma unroll
                for (int k = 0; k < NF; ++k) a[k] += __shfl_xor(a[k], o, 64);
            if (live && (sub & (REM - 1)) == 0) {
#pragma unroll
                for (int k = 0; k < NF; ++k) stf(dxr + cw * g.x_ld + c0 + k, a[k]);
            }
        }
    }
}

// ------------------------------------------------------------------ weight gradient
// One lane = one fine output voxel at a time (grid-stride over slabs), NW = taps * Ci * Co accumulators (+ Co for the bias) in
// registers; lanes are combined by shuffle trees and the four waves in a fixed order, one partial per workgroup, then
// upconv_wgrad_reduce_kernel (double).
template <typename T, int CI, int CO, int NT, int LS>
__global__ void __launch_bounds__(256)
upconv_wgrad_kernel(Mri3dConvGeom g, const T* __restrict__ x, const T* __restrict__ dy, float* __restrict__ part, int hch) {
    constexpr int NW = NT * CI * CO;
    static_assert(NW <= kUcMaxW, "too many accumulators per lane");
    __shared__ float wred[4][NW + CO];
    const UpConvTaps tp = upconv_taps(g);
    const int dc = g.di >> LS, hc = g.hi >> LS, wc = g.wi >> LS;
    float acc[NW], bacc[CO];
#pragma unroll
    for (int i = 0; i < NW; ++i) acc[i] = 0.f;
#pragma unroll
    for (int j = 0; j < CO; ++j) bacc[j] = 0.f;
    const int hchunks = (g.ho + hch - 1) / hch;
    const int slabs = g.n * g.dout * hchunks;
    for (int slab = blockIdx.x; slab < slabs; slab += gridDim.x) {
        const int hk = slab % hchunks, nd = slab / hchunks;
        const int n = nd / g.dout, od = nd - n * g.dout;
        const int h0 = hk * hch, hn = min(hch, g.ho - h0);
        const unsigned inner = (unsigned)hn * g.wo;
        const T* xn = x + (int64_t)n * dc * hc * wc * g.x_ld;
        const T* dyn = dy + (((int64_t)nd * g.ho + h0) * g.wo) * g.y_ld;
        for (unsigned e = threadIdx.x; e < inner; e += 256) {
            const int ow = e % (unsigned)g.wo, oh = h0 + e / (unsigned)g.wo;
            float d[CO];
#pragma unroll
            for (int j = 0; j < CO; ++j) {
                d[j] = ldf(dyn + (int64_t)e * g.y_ld + j);
                bacc[j] += d[j];
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int id = od + tp.od[t], ih = oh + tp.oh[t], iw = ow + tp.ow[t];
                const bool ok = (unsigned)id < (unsigned)g.di && (unsigned)ih < (unsigned)g.hi && (unsigned)iw < (unsigned)g.wi;
                const int off = ok ? (((id >> LS) * hc + (ih >> LS)) * wc + (iw >> LS)) * g.x_ld : 0;
#pragma unroll
                for (int ci = 0; ci < CI; ci += 4) {
                    float4 q = ldf4(xn + off + ci);
                    if (!ok) q = make_float4(0.f, 0.f, 0.f, 0.f);
                    const float qv[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                    for (int k = 0; k < 4; ++k)
#pragma unroll
                        for (int j = 0; j < CO; ++j) acc[(t * CI + ci + k) * CO + j] = fmaf(qv[k], d[j], acc[(t * CI + ci + k) * CO + j]);
                }
            }
        }
    }
    // block sums: an xor-shuffle tree inside each wave, then the four waves in a fixed order (the sum does not depend on scheduling)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NW + CO; ++i) {
        float v = i < NW ? acc[i < NW ? i : 0] : bacc[i < NW ? 0 : i - NW];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane == 0) wred[wv][i] = v;
    }
    __syncthreads();
    float* out = part + (size_t)blockIdx.x * (NW + CO);
    for (int i = threadIdx.x; i < NW + CO; i += 256) out[i] = (wred[0][i] + wred[1][i]) + (wred[2][i] + wred[3][i]);
}

// dw[co][ci][tap] = sum_b part[b][(tap * Ci + ci) * Co + co];  dbias[co] = sum_b part[b][NW + co]   (double, fixed order)
__global__ void __launch_bounds__(64)
upconv_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, float* __restrict__ dbias, int nb, int taps,
                           int Ci, int Co) {
    const int nw = taps * Ci * Co;
    const int i = blockIdx.x;   // element of the partial layout
    double s = 0.0;
    for (int b = threadIdx.x; b < nb; b += 64) s += (double)part[(size_t)b * (nw + Co) + i];
    __shared__ double red[64];
    red[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int k = 0; k < 64; ++k) t += red[k];
        if (i < nw) {
            const int co = i % Co, ci = (i / Co) % Ci, tap = i / (Co * Ci);
            dw[((size_t)co * Ci + ci) * taps + tap] = (float)t;
        } else if (dbias != nullptr) {
            dbias[i - nw] = (float)t;
        }
    }
}

// (Ci, Co, taps) the weight gradient is instantiated for: the shipped autoencoders' last UpBlock (8 -> 1, 3 taps) and its
// neighbours; every other shape keeps the unfused pair of kernels (ops.upsample_conv3d falls back)
#define MRI3D_UPCONV_INSTANCES(X) X(8, 1, 3) X(8, 2, 3) X(4, 1, 3) X(4, 2, 3) X(4, 4, 3) X(16, 1, 3) X(8, 1, 6) X(4, 1, 6) X(4, 2, 6) \
    X(8, 1, 1) X(8, 8, 1) X(4, 4, 1) X(16, 4, 1) X(32, 2, 1) X(4, 1, 1)
static bool upconv_instance(int ci, int co, int taps) {
#define MRI3D_X(CIv, COv, NTv) if (ci == CIv && co == COv && taps == NTv) return true;
    MRI3D_UPCONV_INSTANCES(MRI3D_X)
#undef MRI3D_X
    return false;
}

static bool upconv_ok(const Mri3dConvGeom& g, int S) {
    const int taps = g.kd * g.kh * g.kw;
    return (S == 2 || S == 4) && upconv_instance(g.ci, g.co, taps) && g.sd == 1 && g.sh == 1 && g.sw == 1 && g.dd == 1 && g.dh == 1 && g.dw == 1 && taps >= 1 && taps <= 8 &&
           g.ci % 4 == 0 && g.ci <= 32 && g.co >= 1 && g.co <= 8 && taps * g.ci * g.co <= kUcMaxW && g.x_ld % 4 == 0 && g.x_ld >= g.ci &&
           g.y_ld >= g.co && g.di % S == 0 && g.hi % S == 0 && g.wi % S == 0 && g.n > 0 &&
           g.dout == g.di + 2 * g.pd - g.kd + 1 && g.ho == g.hi + 2 * g.ph - g.kh + 1 && g.wo == g.wi + 2 * g.pw - g.kw + 1 &&
           g.dout > 0 && g.ho > 0 && g.wo > 0 && (int64_t)g.dout * g.ho * g.wo < 0x7fffffff &&
           (int64_t)(g.di / S) * (g.hi / S) * (g.wi / S) * g.x_ld < 0x7fffffff &&
           (g.dtype == MRI3D_F32 || (g.dtype == MRI3D_BF16 && g.x_ld % 4 == 0));
}

static int upconv_hch(const Mri3dConvGeom& g) { return std::max(1, std::min(g.ho, 1024 / std::max(g.wo, 1))); }

}  // namespace mri3d

using namespace mri3d;

extern "C" int32_t mri3d_upconv3d_supported(const Mri3dConvGeom* g, int32_t scale) { return g != nullptr && upconv_ok(*g, scale) ? 1 : 0; }

extern "C" size_t mri3d_upconv3d_workspace_bytes(const Mri3dConvGeom* g, int32_t scale) {
    if (!g || !upconv_ok(*g, scale)) return 0;
    return (size_t)kUcBlocks * (g->kd * g->kh * g->kw * g->ci * g->co + g->co) * sizeof(float);
}

extern "C" int mri3d_upconv3d_fwd(const Mri3dConvGeom* g, int32_t scale, const void* x, const float* w, const float* bias, void* y,
                                  mri3d_stream_t stream) {
    MRI3D_REQUIRE(g && x && w && y, MRI3D_EINVAL, "upconv3d_fwd: null pointer");
    MRI3D_REQUIRE(upconv_ok(*g, scale), MRI3D_ENOTSUP, "upconv3d_fwd: geometry not served (see mri3d_upconv3d_supported)");
    MRI3D_REQUIRE(aligned_vec4(g->dtype, x), MRI3D_EINVAL, "upconv3d_fwd: x must be aligned to four channels");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int hch = upconv_hch(*g);
    const int64_t slabs = (int64_t)g->n * g->dout * cdiv(g->ho, hch);
    const int grid = (int)std::min<int64_t>(slabs, 256 * 16);
    const int taps = g->kd * g->kh * g->kw;
    bool launched = false;
#define MRI3D_UCF(Tv, CIv, COv, NTv)                                                                                    \
    if (!launched && g->ci == CIv && g->co == COv && taps == NTv) {                                                   \
        if (scale == 4)                                                                                               \
            hipLaunchKernelGGL((upconv_fwd_kernel<Tv, CIv, COv, NTv, 2>), dim3(grid), dim3(256), 0, s, *g, (const Tv*)x, w, bias, (Tv*)y, hch); \
        else                                                                                                          \
            hipLaunchKernelGGL((upconv_fwd_kernel<Tv, CIv, COv, NTv, 1>), dim3(grid), dim3(256), 0, s, *g, (const Tv*)x, w, bias, (Tv*)y, hch); \
        launched = true;                                                                                              \
    }
#define MRI3D_X(CIv, COv, NTv) MRI3D_UCF(T, CIv, COv, NTv)
    MRI3D_DISPATCH_DTYPE(g->dtype, T, { MRI3D_UPCONV_INSTANCES(MRI3D_X) });
#undef MRI3D_X
#undef MRI3D_UCF
    MRI3D_REQUIRE(launched, MRI3D_ENOTSUP, "upconv3d_fwd: no kernel instance for Ci %d, Co %d, %d taps", g->ci, g->co, taps);
    return check_launch("upconv3d_fwd");
}

extern "C" int mri3d_upconv3d_dgrad(const Mri3dConvGeom* g, int32_t scale, const void* dy, const float* w, void* dx,
                                    mri3d_stream_t stream) {
    MRI3D_REQUIRE(g && dy && w && dx, MRI3D_EINVAL, "upconv3d_dgrad: null pointer");
    MRI3D_REQUIRE(upconv_ok(*g, scale), MRI3D_ENOTSUP, "upconv3d_dgrad: geometry not served (see mri3d_upconv3d_supported)");
    MRI3D_REQUIRE(aligned_vec4(g->dtype, dx), MRI3D_EINVAL, "upconv3d_dgrad: dx must be aligned to four channels");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int64_t rows = (int64_t)g->n * (g->di / scale) * (g->hi / scale);
    MRI3D_REQUIRE(rows <= 0x7fffffff, MRI3D_ENOTSUP, "upconv3d_dgrad: too many coarse rows");
    const int grid = (int)std::min<int64_t>(rows, 256 * 32);
    const int taps = g->kd * g->kh * g->kw;
    bool launched = false;
#define MRI3D_UCD(Tv, CIv, COv, NTv)                                                                                    \
    if (!launched && g->ci == CIv && g->co == COv && taps == NTv) {                                                   \
        if (scale == 4)                                                                                               \
            hipLaunchKernelGGL((upconv_dgrad_kernel<Tv, CIv, COv, NTv, 2>), dim3(grid), dim3(256), 0, s, *g, (const Tv*)dy, w, (Tv*)dx, (int)rows); \
        else                                                                                                          \
            hipLaunchKernelGGL((upconv_dgrad_kernel<Tv, CIv, COv, NTv, 1>), dim3(grid), dim3(256), 0, s, *g, (const Tv*)dy, w, (Tv*)dx, (int)rows); \
        launched = true;                                                                                              \
    }
#define MRI3D_X(CIv, COv, NTv) MRI3D_UCD(T, CIv, COv, NTv)
    MRI3D_DISPATCH_DTYPE(g->dtype, T, { MRI3D_UPCONV_INSTANCES(MRI3D_X) });
#undef MRI3D_X
#undef MRI3D_UCD
    MRI3D_REQUIRE(launched, MRI3D_ENOTSUP, "upconv3d_dgrad: no kernel instance for Ci %d, Co %d, %d taps", g->ci, g->co, taps);
    return check_launch("upconv3d_dgrad");
}

extern "C" int mri3d_upconv3d_wgrad(const Mri3dConvGeom* g, int32_t scale, const void* x, const void* dy, float* dw, float* dbias,
                                    void* workspace, size_t ws_bytes, mri3d_stream_t stream) {
    MRI3D_REQUIRE(g && x && dy && dw, MRI3D_EINVAL, "upconv3d_wgrad: null pointer");
    MRI3D_REQUIRE(upconv_ok(*g, scale), MRI3D_ENOTSUP, "upconv3d_wgrad: geometry not served (see mri3d_upconv3d_supported)");
    MRI3D_REQUIRE(aligned_vec4(g->dtype, x), MRI3D_EINVAL, "upconv3d_wgrad: x must be aligned to four channels");
    MRI3D_REQUIRE(workspace && ws_bytes >= mri3d_upconv3d_workspace_bytes(g, scale), MRI3D_EINVAL, "upconv3d_wgrad: workspace too small");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int taps = g->kd * g->kh * g->kw;
    const int hch = upconv_hch(*g);
    const int64_t slabs = (int64_t)g->n * g->dout * cdiv(g->ho, hch);
    const int nb = (int)std::min<int64_t>(slabs, kUcBlocks);
    float* part = static_cast<float*>(workspace);
    bool launched = false;
#define MRI3D_UCW(Tv, CIv, COv, NTv)                                                                                  \
    if (!launched && g->ci == CIv && g->co == COv && taps == NTv) {                                                   \
        if (scale == 4)                                                                                               \
            hipLaunchKernelGGL((upconv_wgrad_kernel<Tv, CIv, COv, NTv, 2>), dim3(nb), dim3(256), 0, s, *g, (const Tv*)x, \
                               (const Tv*)dy, part, hch);                                                             \
        else                                                                                                          \
            hipLaunchKernelGGL((upconv_wgrad_kernel<Tv, CIv, COv, NTv, 1>), dim3(nb), dim3(256), 0, s, *g, (const Tv*)x, \
                               (const Tv*)dy, part, hch);                                                             \
        launched = true;                                                                                              \
    }
#define MRI3D_X(CIv, COv, NTv) MRI3D_UCW(T, CIv, COv, NTv)
    MRI3D_DISPATCH_DTYPE(g->dtype, T, { MRI3D_UPCONV_INSTANCES(MRI3D_X) });
#undef MRI3D_X
#undef MRI3D_UCW
    MRI3D_REQUIRE(launched, MRI3D_ENOTSUP, "upconv3d_wgrad: no kernel instance for Ci %d, Co %d, %d taps", g->ci, g->co, taps);
    int rc = check_launch("upconv3d_wgrad");
    if (rc) return rc;
    hipLaunchKernelGGL(upconv_wgrad_reduce_kernel, dim3(taps * g->ci * g->co + g->co), dim3(64), 0, s, part, dw, dbias, nb, taps,
                       g->ci, g->co);
    return check_launch("upconv3d_wgrad(reduce)");
}
