// resample.hip — MaxPool3d (fwd with 1-byte arg-max, bwd gather) and Upsample (nearest / trilinear, fwd gather,
// bwd transposed gather), NDHWC with voxel pitch.  Reference call sites: nn.MaxPool3d(2) in unet.UNet encoder,
// AE_model.py:27, cnn_model.py:115-148; MaxPool3d(4,2) cnn_model.py:221,232; nn.Upsample(trilinear) in unet.UNet
// decoder; nearest in modified_3dunet.py:13 and AE_model.py:70-73; F.interpolate(size=) AE_model.py:119.
//
// All four kernels are HBM-bound streaming passes (algorithmic bytes = one read of the source + one write of the
// destination); lanes run along channels then voxels so every wave touches contiguous NDHWC memory.  Backward passes
// are written as gathers (each destination element is produced by exactly one thread) => deterministic, no atomics.
#include "common.h"
#include <limits.h>
#include <algorithm>
#include <stdlib.h>

namespace mri3d {

template <int VEC>
struct V {
    float v[VEC];
    template <typename T>
    __device__ __forceinline__ void load(const T* p) {
        if constexpr (VEC == 8) {   // bf16 only: one 16-byte access (8-byte ones leave these kernels at half the fp32 rate)
            const bf16x8_t t = *reinterpret_cast<const bf16x8_t*>(p);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (float)t[j];
        } else if constexpr (VEC == 4) {
            float4 t = ldf4(p);
            v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
        } else {
            v[0] = ldf(p);
        }
    }
    template <typename T>
    __device__ __forceinline__ void store(T* p) const {
        if constexpr (VEC == 8) {
            bf16x8_t o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (bf16_t)v[j];
            *reinterpret_cast<bf16x8_t*>(p) = o;
        } else if constexpr (VEC == 4) stf4(p, make_float4(v[0], v[1], v[2], v[3]));
        else stf(p, v[0]);
    }
};

// Indexing scheme of every kernel in this file: a block walks (n, depth) slabs (scalar decode, 64-bit safe) and its
// threads walk the (h, w, channel-vector) elements of the slab with 32-bit arithmetic — 64-bit divisions per element
// made the first version of these kernels VALU-bound at 1.3-2.2 TB/s.

// ------------------------------------------------------------------ max pool forward
template <typename T, int VEC>
__global__ void __launch_bounds__(256)
maxpool_fwd_kernel(Mri3dPoolGeom g, const T* __restrict__ x, T* __restrict__ y, uint8_t* __restrict__ idx, int hch) {
    const unsigned CV = g.c / VEC;
    const int hchunks = (g.ho + hch - 1) / hch;
    const int slabs = g.n * g.dout * hchunks;
    for (int slab = blockIdx.x; slab < slabs; slab += gridDim.x) {
        const int hc = slab % hchunks, nd = slab / hchunks;
        const int n = nd / g.dout, od = nd - n * g.dout;
        const int h0 = hc * hch, hn = min(hch, g.ho - h0);
        const unsigned inner = (unsigned)hn * g.wo * CV;
        const T* xn = x + (int64_t)n * g.di * g.hi * g.wi * g.x_ld;
        const int64_t obase = ((int64_t)nd * g.ho + h0) * g.wo;
        for (unsigned e = threadIdx.x; e < inner; e += blockDim.x) {
            const unsigned cv = e % CV, pix = e / CV;
            const int ow = pix % g.wo, oh = h0 + pix / g.wo;
            float best[VEC];
            int bi[VEC];
#pragma unroll
            for (int j = 0; j < VEC; ++j) { best[j] = -INFINITY; bi[j] = 0; }
            bool first = true;
            for (int kd = 0; kd < g.kd; ++kd) {
                const int id = od * g.sd - g.pd + kd;
                if ((unsigned)id >= (unsigned)g.di) continue;
                for (int kh = 0; kh < g.kh; ++kh) {
                    const int ih = oh * g.sh - g.ph + kh;
                    if ((unsigned)ih >= (unsigned)g.hi) continue;
                    for (int kw = 0; kw < g.kw; ++kw) {
                        const int iw = ow * g.sw - g.pw + kw;
                        if ((unsigned)iw >= (unsigned)g.wi) continue;
                        V<VEC> xv;
                        xv.load(xn + (((int64_t)id * g.hi + ih) * g.wi + iw) * g.x_ld + cv * VEC);
                        const int tap = (kd * g.kh + kh) * g.kw + kw;
#pragma unroll
                        for (int j = 0; j < VEC; ++j) {
                            // torch: first maximum in raster order wins; NaN propagates
                            if (first || xv.v[j] > best[j] || xv.v[j] != xv.v[j]) {
                                best[j] = xv.v[j];
                                bi[j] = tap;
                            }
                        }
                        first = false;
                    }
                }
            }
            V<VEC> o;
#pragma unroll
            for (int j = 0; j < VEC; ++j) o.v[j] = best[j];
            const int64_t ov = obase + pix;
            o.store(y + ov * g.y_ld + cv * VEC);
            uint8_t* ip = idx + ov * g.c + cv * VEC;
            if constexpr (VEC == 8) {
                uint2 pk;
                pk.x = (uint32_t)bi[0] | ((uint32_t)bi[1] << 8) | ((uint32_t)bi[2] << 16) | ((uint32_t)bi[3] << 24);
                pk.y = (uint32_t)bi[4] | ((uint32_t)bi[5] << 8) | ((uint32_t)bi[6] << 16) | ((uint32_t)bi[7] << 24);
                *reinterpret_cast<uint2*>(ip) = pk;
            } else if constexpr (VEC == 4) {
                *reinterpret_cast<uint32_t*>(ip) = (uint32_t)bi[0] | ((uint32_t)bi[1] << 8) |
                                                   ((uint32_t)bi[VEC > 2 ? 2 : 0] << 16) | ((uint32_t)bi[VEC > 3 ? 3 : 0] << 24);
            } else {
                ip[0] = (uint8_t)bi[0];
            }
        }
    }
}

// ------------------------------------------------------------------ max pool backward (gather over covering windows)
template <typename T, int VEC>
__global__ void __launch_bounds__(256)
maxpool_bwd_kernel(Mri3dPoolGeom g, const T* __restrict__ dy, const uint8_t* __restrict__ idx,
                   T* __restrict__ dx, int hch, const T* __restrict__ addend, int a_ld) {
    // addend (optional): a second gradient of the pooled tensor's INPUT, summed in here — the encoder output feeds both the
    // pool and the skip connection, and autograd would otherwise add the two gradients in a separate full-resolution pass
    const unsigned CV = g.c / VEC;
    const int hchunks = (g.hi + hch - 1) / hch;
    const int slabs = g.n * g.di * hchunks;
    for (int slab = blockIdx.x; slab < slabs; slab += gridDim.x) {
        const int hc = slab % hchunks, nd = slab / hchunks;
        const int n = nd / g.di, id = nd - n * g.di;
        const int h0 = hc * hch, hn = min(hch, g.hi - h0);
        const unsigned inner = (unsigned)hn * g.wi * CV;
        // windows od with od*sd - pd <= id <= od*sd - pd + kd - 1
        int od_lo = id + g.pd - g.kd + 1; od_lo = od_lo <= 0 ? 0 : (od_lo + g.sd - 1) / g.sd;
        int od_hi = (id + g.pd) / g.sd; if (od_hi > g.dout - 1) od_hi = g.dout - 1;
        const int64_t ibase = ((int64_t)nd * g.hi + h0) * g.wi;
        for (unsigned e = threadIdx.x; e < inner; e += blockDim.x) {
            const unsigned cv = e % CV, pix = e / CV;
            const int iw = pix % g.wi, ih = h0 + pix / g.wi;
            float acc[VEC];
#pragma unroll
            for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
            if (addend != nullptr) {
                V<VEC> av;
                av.load(addend + (ibase + pix) * a_ld + cv * VEC);
#pragma unroll
                for (int j = 0; j < VEC; ++j) acc[j] = av.v[j];
            }
            int oh_lo = ih + g.ph - g.kh + 1; oh_lo = oh_lo <= 0 ? 0 : (oh_lo + g.sh - 1) / g.sh;
            int oh_hi = (ih + g.ph) / g.sh; if (oh_hi > g.ho - 1) oh_hi = g.ho - 1;
            int ow_lo = iw + g.pw - g.kw + 1; ow_lo = ow_lo <= 0 ? 0 : (ow_lo + g.sw - 1) / g.sw;
            int ow_hi = (iw + g.pw) / g.sw; if (ow_hi > g.wo - 1) ow_hi = g.wo - 1;
            for (int od = od_lo; od <= od_hi; ++od) {
                const int kd = id - (od * g.sd - g.pd);
                for (int oh = oh_lo; oh <= oh_hi; ++oh) {
                    const int kh = ih - (oh * g.sh - g.ph);
                    for (int ow = ow_lo; ow <= ow_hi; ++ow) {
                        const int kw = iw - (ow * g.sw - g.pw);
                        const int tap = (kd * g.kh + kh) * g.kw + kw;
                        const int64_t ov = (((int64_t)n * g.dout + od) * g.ho + oh) * g.wo + ow;
                        V<VEC> gv;
                        gv.load(dy + ov * g.y_ld + cv * VEC);
                        const uint8_t* ip = idx + ov * g.c + cv * VEC;
                        if constexpr (VEC == 8) {
                            const uint2 pk = *reinterpret_cast<const uint2*>(ip);
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                if ((int)((pk.x >> (8 * j)) & 0xff) == tap) acc[j] += gv.v[j];
                                if ((int)((pk.y >> (8 * j)) & 0xff) == tap) acc[4 + j] += gv.v[4 + j];
                            }
                        } else if constexpr (VEC == 4) {
                            const uint32_t pk = *reinterpret_cast<const uint32_t*>(ip);
#pragma unroll
                            for (int j = 0; j < VEC; ++j)
                                if ((int)((pk >> (8 * j)) & 0xff) == tap) acc[j] += gv.v[j];
                        } else {
                            if ((int)ip[0] == tap) acc[0] += gv.v[0];
                        }
                    }
                }
            }
            V<VEC> o;
#pragma unroll
            for (int j = 0; j < VEC; ++j) o.v[j] = acc[j];
            o.store(dx + (ibase + pix) * g.x_ld + cv * VEC);
        }
    }
}

// ------------------------------------------------------------------ MaxPool3d(2) forward, line-contiguous lanes
// kernel 2 / stride 2 / no padding on even extents (every pooling of the reference: cnn_model.py:115-148, unet.UNet).  In the slab
// kernel above a lane owns an OUTPUT voxel: consecutive lanes read input voxels two apart, so every wave-load touches half of
// each 128-byte line and the kw = 1 tap comes back for the other half.  Here a lane owns one INPUT column (2*ow + kw) x VEC
// channels: consecutive lanes read consecutive bytes (whole lines per wave-load), take the maximum over their four (kd, kh)
// taps, and the two kw lanes of an output voxel — CV lanes apart — are merged with one shuffle.  Same rule as the slab kernel
// (first maximum in raster order wins, NaN propagates: the later NaN's index), same index bytes.
template <typename T, int VEC>
__global__ void __launch_bounds__(256)
maxpool2_fwd_kernel(Mri3dPoolGeom g, const T* __restrict__ x, T* __restrict__ y, uint8_t* __restrict__ idx, int hch) {
    const unsigned CV = g.c / VEC;            // lanes per voxel: a power of two <= 32 (host)
    const int hchunks = (g.ho + hch - 1) / hch;
    const int slabs = g.n * g.dout * hchunks;
    const unsigned row = (unsigned)g.wi * g.x_ld, plane = (unsigned)g.hi * row;   // element strides of the input (one sample < 2^31)
    for (int slab = blockIdx.x; slab < slabs; slab += gridDim.x) {
        const int hc = slab % hchunks, nd = slab / hchunks;
        const int n = nd / g.dout, od = nd - n * g.dout;
        const int h0 = hc * hch, hn = min(hch, g.ho - h0);
        const unsigned inner = (unsigned)hn * g.wi * CV;   // lanes walk (oh, input column iw, channel vector)
        const T* xs = x + ((int64_t)n * g.di + 2 * od) * g.hi * g.wi * g.x_ld;
        const int64_t obase = ((int64_t)nd * g.ho + h0) * g.wo;
        for (unsigned e0 = 0; e0 < inner; e0 += blockDim.x) {   // whole block iterates together: the shuffle needs both kw lanes
            const unsigned e = e0 + threadIdx.x;
            const bool live = e < inner;
            const unsigned ee = live ? e : 0;
            const unsigned cv = ee % CV, col = ee / CV;
            const unsigned iw = col % g.wi, ohl = col / g.wi;
            const unsigned kw = iw & 1, ow = iw >> 1, oh = h0 + ohl;
            const T* p0 = xs + (2 * oh) * row + iw * (unsigned)g.x_ld + cv * VEC;
            V<VEC> v[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) v[t].load(p0 + (t >> 1) * plane + (t & 1) * row);
            float best[VEC];
            int bi[VEC];
#pragma unroll
            for (int j = 0; j < VEC; ++j) { best[j] = v[0].v[j]; bi[j] = (int)kw; }
#pragma unroll
            for (int t = 1; t < 4; ++t)
#pragma unroll
                for (int j = 0; j < VEC; ++j)
                    if (v[t].v[j] > best[j] || v[t].v[j] != v[t].v[j]) { best[j] = v[t].v[j]; bi[j] = 2 * t + (int)kw; }
            // merge with the other kw lane (CV lanes away; wi is even and CV | 64, so both are in the same wave and both live)
            V<VEC> o;
            int oi[VEC];
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                const float pb = __shfl_xor(best[j], (int)CV, 64);
                const int pi = __shfl_xor(bi[j], (int)CV, 64);
                const bool mine_nan = best[j] != best[j], his_nan = pb != pb;
                bool take_his;
                if (mine_nan || his_nan) take_his = his_nan && (!mine_nan || pi > bi[j]);     // the LAST NaN in raster order
                else take_his = pb > best[j] || (pb == best[j] && pi < bi[j]);                  // the FIRST maximum
                o.v[j] = take_his ? pb : best[j];
                oi[j] = take_his ? pi : bi[j];
            }
            if (live && kw == 0) {
                const int64_t ov = obase + (int64_t)ohl * g.wo + ow;
                o.store(y + ov * g.y_ld + cv * VEC);
                uint8_t* ip = idx + ov * g.c + cv * VEC;
                if constexpr (VEC == 8) {
                    uint2 pk;
                    pk.x = (uint32_t)oi[0] | ((uint32_t)oi[1] << 8) | ((uint32_t)oi[2] << 16) | ((uint32_t)oi[3] << 24);
                    pk.y = (uint32_t)oi[4] | ((uint32_t)oi[5] << 8) | ((uint32_t)oi[6] << 16) | ((uint32_t)oi[7] << 24);
                    *reinterpret_cast<uint2*>(ip) = pk;
                } else {
                    *reinterpret_cast<uint32_t*>(ip) = (uint32_t)oi[0] | ((uint32_t)oi[1] << 8) | ((uint32_t)oi[2] << 16) | ((uint32_t)oi[3] << 24);
                }
            }
        }
    }
}

static inline bool pool2_ok(const Mri3dPoolGeom& g, int vec) {
    const int cv = g.c / vec;
    return g.kd == 2 && g.kh == 2 && g.kw == 2 && g.sd == 2 && g.sh == 2 && g.sw == 2 && g.pd == 0 && g.ph == 0 && g.pw == 0 &&
           g.di == 2 * g.dout && g.hi == 2 * g.ho && g.wi == 2 * g.wo && cv >= 1 && cv <= 32 && (cv & (cv - 1)) == 0 &&
           (int64_t)g.di * g.hi * g.wi * g.x_ld < ((int64_t)1 << 31);
}

// ------------------------------------------------------------------ upsample coordinate helpers (torch semantics)
struct Lin { int i0, i1; float l0, l1; };
__device__ __forceinline__ Lin lin_src(int o, float r, int in_size, int align_corners) {
    // ATen area_pixel_compute_source_index + guard index (UpSample.h)
    float src = align_corners ? r * (float)o : fmaf(r, (float)o + 0.5f, -0.5f);
    if (!align_corners && src < 0.f) src = 0.f;
    Lin L;
    L.i0 = (int)src;
    if (L.i0 > in_size - 1) L.i0 = in_size - 1;
    L.i1 = L.i0 + (L.i0 < in_size - 1 ? 1 : 0);
    L.l1 = src - (float)L.i0;
    if (L.l1 < 0.f) L.l1 = 0.f;
    if (L.l1 > 1.f) L.l1 = 1.f;
    L.l0 = 1.f - L.l1;
    return L;
}
__device__ __forceinline__ int near_src(int o, float r, int in_size) {
    int i = (int)floorf((float)o * r);
    return i < in_size - 1 ? i : in_size - 1;
}

template <typename T, int VEC>
__global__ void __launch_bounds__(256)
upsample_fwd_kernel(Mri3dUpGeom g, const T* __restrict__ x, T* __restrict__ y, int hch) {
    const unsigned CV = g.c / VEC;
    const int hchunks = (g.ho + hch - 1) / hch;
    const int slabs = g.n * g.dout * hchunks;
    for (int slab = blockIdx.x; slab < slabs; slab += gridDim.x) {
        const int hc = slab % hchunks, nd = slab / hchunks;
        const int n = nd / g.dout, od = nd - n * g.dout;
        const int h0 = hc * hch, hn = min(hch, g.ho - h0);
        const unsigned inner = (unsigned)hn * g.wo * CV;
        const T* xn = x + (int64_t)n * g.di * g.hi * g.wi * g.x_ld;
        const int64_t obase = ((int64_t)nd * g.ho + h0) * g.wo;
        // depth taps are uniform over the slab
        int idn = 0;
        Lin Ld_ = {0, 0, 1.f, 0.f};
        if (g.mode == MRI3D_UP_NEAREST) idn = near_src(od, g.rd, g.di);
        else Ld_ = lin_src(od, g.rd, g.di, g.align_corners);
        for (unsigned e = threadIdx.x; e < inner; e += blockDim.x) {
            const unsigned cv = e % CV, pix = e / CV;
            const int ow = pix % g.wo, oh = h0 + pix / g.wo;
            const T* xc = xn + cv * VEC;
            V<VEC> o;
            if (g.mode == MRI3D_UP_NEAREST) {
                const int ih = near_src(oh, g.rh, g.hi), iw = near_src(ow, g.rw, g.wi);
                o.load(xc + (((int64_t)idn * g.hi + ih) * g.wi + iw) * g.x_ld);
            } else {
                const Lin Lh = lin_src(oh, g.rh, g.hi, g.align_corners);
                const Lin Lw = lin_src(ow, g.rw, g.wi, g.align_corners);
#pragma unroll
                for (int j = 0; j < VEC; ++j) o.v[j] = 0.f;
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const int id = a ? Ld_.i1 : Ld_.i0;
                    const float wd = a ? Ld_.l1 : Ld_.l0;
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        const int ih = b ? Lh.i1 : Lh.i0;
                        const float wh = wd * (b ? Lh.l1 : Lh.l0);
#pragma unroll
                        for (int c = 0; c < 2; ++c) {
                            const int iw = c ? Lw.i1 : Lw.i0;
                            const float w = wh * (c ? Lw.l1 : Lw.l0);
                            V<VEC> xv;
                            xv.load(xc + (((int64_t)id * g.hi + ih) * g.wi + iw) * g.x_ld);
#pragma unroll
                            for (int j = 0; j < VEC; ++j) o.v[j] = fmaf(w, xv.v[j], o.v[j]);
                        }
                    }
                }
            }
            o.store(y + (obase + pix) * g.y_ld + cv * VEC);
        }
    }
}

// Per-axis tables: for every source index i, the [lo, hi] range of destination indices that read it.
// grid = 3 blocks (one per axis); tab layout: [lo_d(di) | hi_d(di) | lo_h(hi) | hi_h(hi) | lo_w(wi) | hi_w(wi)]
__global__ void upsample_tables_kernel(Mri3dUpGeom g, int* __restrict__ tab) {
    int axis = blockIdx.x;
    int in_size = axis == 0 ? g.di : (axis == 1 ? g.hi : g.wi);
    int out_size = axis == 0 ? g.dout : (axis == 1 ? g.ho : g.wo);
    float r = axis == 0 ? g.rd : (axis == 1 ? g.rh : g.rw);
    int* lo = tab + (axis == 0 ? 0 : (axis == 1 ? 2 * g.di : 2 * g.di + 2 * g.hi));
    int* hi = lo + in_size;
    for (int i = threadIdx.x; i < in_size; i += blockDim.x) { lo[i] = INT_MAX; hi[i] = -1; }
    __syncthreads();
    for (int o = threadIdx.x; o < out_size; o += blockDim.x) {
        if (g.mode == MRI3D_UP_NEAREST) {
            int i = near_src(o, r, in_size);
            atomicMin(&lo[i], o);
            atomicMax(&hi[i], o);
        } else {
            Lin L = lin_src(o, r, in_size, g.align_corners);
            atomicMin(&lo[L.i0], o);
            atomicMax(&hi[L.i0], o);
            atomicMin(&lo[L.i1], o);
            atomicMax(&hi[L.i1], o);
        }
    }
}

__device__ __forceinline__ float up_weight(int o, int i, float r, int in_size, int mode, int align_corners) {
    if (mode == MRI3D_UP_NEAREST) return near_src(o, r, in_size) == i ? 1.f : 0.f;
    Lin L = lin_src(o, r, in_size, align_corners);
    return (L.i0 == i ? L.l0 : 0.f) + (L.i1 == i ? L.l1 : 0.f);
}

template <typename T, int VEC>
__global__ void __launch_bounds__(256)
upsample_bwd_kernel(Mri3dUpGeom g, const T* __restrict__ dy, T* __restrict__ dx, const int* __restrict__ tab, int hch) {
    const unsigned CV = g.c / VEC;
    const int* lo_d = tab; const int* hi_d = tab + g.di;
    const int* lo_h = tab + 2 * g.di; const int* hi_h = lo_h + g.hi;
    const int* lo_w = tab + 2 * g.di + 2 * g.hi; const int* hi_w = lo_w + g.wi;
    const int hchunks = (g.hi + hch - 1) / hch;
    const int slabs = g.n * g.di * hchunks;
    for (int slab = blockIdx.x; slab < slabs; slab += gridDim.x) {
        const int hc = slab % hchunks, nd = slab / hchunks;
        const int n = nd / g.di, id = nd - n * g.di;
        const int hh0 = hc * hch, hn = min(hch, g.hi - hh0);
        const unsigned inner = (unsigned)hn * g.wi * CV;
        const int d0 = lo_d[id], d1 = hi_d[id];
        const T* dn = dy + (int64_t)n * g.dout * g.ho * g.wo * g.y_ld;
        const int64_t ibase = ((int64_t)nd * g.hi + hh0) * g.wi;
        for (unsigned e = threadIdx.x; e < inner; e += blockDim.x) {
            const unsigned cv = e % CV, pix = e / CV;
            const int iw = pix % g.wi, ih = hh0 + pix / g.wi;
            float acc[VEC];
#pragma unroll
            for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
            const int h0 = lo_h[ih], h1 = hi_h[ih], w0 = lo_w[iw], w1 = hi_w[iw];
            const T* dc = dn + cv * VEC;
            for (int od = d0; od <= d1; ++od) {
                const float wd = up_weight(od, id, g.rd, g.di, g.mode, g.align_corners);
                if (wd == 0.f) continue;
                for (int oh = h0; oh <= h1; ++oh) {
                    const float wh = wd * up_weight(oh, ih, g.rh, g.hi, g.mode, g.align_corners);
                    if (wh == 0.f) continue;
                    for (int ow = w0; ow <= w1; ++ow) {
                        const float w = wh * up_weight(ow, iw, g.rw, g.wi, g.mode, g.align_corners);
                        if (w == 0.f) continue;
                        V<VEC> gv;
                        gv.load(dc + (((int64_t)od * g.ho + oh) * g.wo + ow) * g.y_ld);
#pragma unroll
                        for (int j = 0; j < VEC; ++j) acc[j] = fmaf(w, gv.v[j], acc[j]);
                    }
                }
            }
            V<VEC> o;
#pragma unroll
            for (int j = 0; j < VEC; ++j) o.v[j] = acc[j];
            o.store(dx + (ibase + pix) * g.x_ld + cv * VEC);
        }
    }
}

// ------------------------------------------------------------------ nearest-neighbour backward with an integer scale
// The decoder of the autoencoder upsamples by 4 with nn.Upsample(mode='nearest') (AE_model.py:110-120): output size = S x input
// size exactly, so source(o) = o / S and the gradient of a coarse voxel is the plain sum of its S x S x S fine voxels.  The generic
// kernel above evaluates three interpolation weights per tap and loads inside `if (w != 0)` (one round trip per tap: 2.9 TB/s on
// the 8-channel layer); here a lane owns (coarse voxel, 4-channel quad) and issues the S x S loads of one fine plane together.
template <typename T, int S>
__global__ void __launch_bounds__(256)
upsample_nearest_int_bwd_kernel(Mri3dUpGeom g, const T* __restrict__ dy, T* __restrict__ dx, int hch) {
    const unsigned CV = g.c / 4;
    const int hchunks = (g.hi + hch - 1) / hch;
    const int slabs = g.n * g.di * hchunks;
    for (int slab = blockIdx.x; slab < slabs; slab += gridDim.x) {
        const int hc = slab % hchunks, nd = slab / hchunks;
        const int n = nd / g.di, id = nd - n * g.di;
        const int hh0 = hc * hch, hn = min(hch, g.hi - hh0);
        const unsigned inner = (unsigned)hn * g.wi * CV;
        const T* dn = dy + ((int64_t)n * g.dout + (int64_t)id * S) * g.ho * g.wo * g.y_ld;   // the coarse plane's first fine plane
        const int64_t ibase = ((int64_t)nd * g.hi + hh0) * g.wi;
        for (unsigned e = threadIdx.x; e < inner; e += blockDim.x) {
            const unsigned cv = e % CV, pix = e / CV;
            const int iw = pix % g.wi, ih = hh0 + pix / g.wi;
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
            const T* dc = dn + (((int64_t)ih * S) * g.wo + (int64_t)iw * S) * g.y_ld + cv * 4;
#pragma unroll 1
            for (int a = 0; a < S; ++a) {
                V<4> gv[S][S];
#pragma unroll
                for (int b = 0; b < S; ++b)
#pragma unroll
                    for (int c = 0; c < S; ++c) gv[b][c].load(dc + (((int64_t)a * g.ho + b) * g.wo + c) * g.y_ld);
#pragma unroll
                for (int b = 0; b < S; ++b)
#pragma unroll
                    for (int c = 0; c < S; ++c)
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[j] += gv[b][c].v[j];
            }
            V<4> o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o.v[j] = acc[j];
            o.store(dx + (ibase + pix) * g.x_ld + cv * 4);
        }
    }
}

// ------------------------------------------------------------------ trilinear x2 backward, LDS-tiled
// The decoder's nn.Upsample(scale_factor=2, mode='trilinear', align_corners=False) (unet.UNet) is the
// only interpolating resample on the hot path, and its transposed gather reads 4x4x4 fine voxels per coarse voxel while
// every fine voxel is shared by 2x2x2 coarse ones: the generic kernel above issues 64 global loads per 16-byte result
// (1.0 TB/s measured).  Here a workgroup stages the (2*2+2) x (2*4+2) x (2*16+2) fine halo of a 2x4x16 coarse tile for
// 8 channels in LDS as fp32 (one coalesced pass, each fine voxel read ~2x from L2 instead of 8x), and every lane gathers
// its 64 taps from LDS with immediate offsets.  Per-axis weights of the 4 taps o = 2i-1 .. 2i+2 are (.25,.75,.75,.25),
// except at the borders where ATen's clamped source index folds a whole fine voxel into the edge: i = 0 -> (0,1,.75,.25),
// i = S-1 -> (.25,.75,1,0).
__device__ __forceinline__ void up2_axis_weights(int i, int S, float (&w)[4]) {
    w[0] = 0.25f; w[1] = 0.75f; w[2] = 0.75f; w[3] = 0.25f;
    if (i == 0) { w[0] = 0.f; w[1] = 1.f; }
    if (i == S - 1) { w[2] = 1.f; w[3] = 0.f; }
}

// CQ = channel quads per pass (2: 8 channels, 4: 16 channels).  A voxel's slice of CQ*4 channels is what one staging
// request reads contiguously: 32 B (CQ = 2) left three quarters of every 128-byte line unused (2 TB/s of L2->LDS traffic
// measured as the bound), so the 16-channel variant with a 2x2x16 coarse tile is used whenever C % 16 == 0.
template <int CQ> struct Up2Tile;
template <> struct Up2Tile<2> { static constexpr int TD = 2, TH = 4, TW = 16; };
template <> struct Up2Tile<4> { static constexpr int TD = 2, TH = 2, TW = 16; };

template <typename T, int CQ>
__global__ void __launch_bounds__(256, 2)
upsample2x_bwd_kernel(Mri3dUpGeom g, const T* __restrict__ dy, T* __restrict__ dx, int tilesD, int tilesH, int tilesW,
                      int ntiles) {
    constexpr int UTD = Up2Tile<CQ>::TD, UTH = Up2Tile<CQ>::TH, UTW = Up2Tile<CQ>::TW;
    constexpr int UFD = 2 * UTD + 2, UFH = 2 * UTH + 2, UFW = 2 * UTW + 2, UFV = UFD * UFH * UFW;
    static_assert(UTD * UTH * UTW * CQ == 256, "one lane per (coarse voxel, channel quad)");
    extern __shared__ __attribute__((aligned(16))) float4 ubuf[];   // [fine voxel][CQ channel quads]
    const int tid = threadIdx.x;
    const int q = tid % CQ, v = tid / CQ;
    const int iwl = v % UTW, ihl = (v / UTW) % UTH, idl = v / (UTW * UTH);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int t = tile;
        const int w0 = (t % tilesW) * UTW;
        t /= tilesW;
        const int h0 = (t % tilesH) * UTH;
        t /= tilesH;
        const int d0 = (t % tilesD) * UTD;
        const int n = t / tilesD;
        const T* dn = dy + (int64_t)n * g.dout * g.ho * g.wo * g.y_ld;
        const int id = d0 + idl, ih = h0 + ihl, iw = w0 + iwl;
        const bool vok = id < g.di && ih < g.hi && iw < g.wi;
        float wd[4], wh[4], ww[4];
        up2_axis_weights(id, g.di, wd);
        up2_axis_weights(ih, g.hi, wh);
        up2_axis_weights(iw, g.wi, ww);
        for (int c0 = 0; c0 < g.c; c0 += 4 * CQ) {
            __syncthreads();
            // pieces are fetched (clamped, unconditional) in batches of 10 before their LDS writes: a load inside a branch
            // serialises the loop into dependent round trips
            constexpr int NP = (UFV * CQ + 255) / 256, NB = 10;
#pragma unroll
            for (int j0 = 0; j0 < NP; j0 += NB) {
                float4 pv[NB];
                unsigned okm = 0;
#pragma unroll
                for (int jj = 0; jj < NB; ++jj) {
                    const int idx = (j0 + jj) * 256 + tid;
                    const int qq = idx % CQ, fv = idx / CQ;
                    const int fw = fv % UFW, t2 = fv / UFW;
                    const int fh = t2 % UFH, fd = t2 / UFH;
                    const int od = 2 * d0 - 1 + fd, oh = 2 * h0 - 1 + fh, ow = 2 * w0 - 1 + fw;
                    const int cc = c0 + 4 * qq;
                    const bool ok = j0 + jj < NP && idx < UFV * CQ && (unsigned)od < (unsigned)g.dout &&
                                    (unsigned)oh < (unsigned)g.ho && (unsigned)ow < (unsigned)g.wo && cc < g.c;
                    okm |= ok ? (1u << jj) : 0u;
                    const int cd = min(max(od, 0), g.dout - 1), ch = min(max(oh, 0), g.ho - 1), cw = min(max(ow, 0), g.wo - 1);
                    pv[jj] = ldf4(dn + (((int64_t)cd * g.ho + ch) * g.wo + cw) * g.y_ld + (cc < g.c ? cc : 0));
                }
#pragma unroll
                for (int jj = 0; jj < NB; ++jj) {
                    const int idx = (j0 + jj) * 256 + tid;
                    const bool ok = (okm >> jj) & 1u;
                    float4 val;
                    val.x = ok ? pv[jj].x : 0.f; val.y = ok ? pv[jj].y : 0.f; val.z = ok ? pv[jj].z : 0.f; val.w = ok ? pv[jj].w : 0.f;
                    if (j0 + jj < NP && idx < UFV * CQ) ubuf[idx] = val;
                }
            }
            __syncthreads();
            if (vok && c0 + 4 * q < g.c) {
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
                const float4* base = ubuf + (((2 * idl) * UFH + 2 * ihl) * UFW + 2 * iwl) * CQ + q;
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        const float wab = wd[a] * wh[b];
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const float w = wab * ww[c];
                            const float4 gv = base[((a * UFH + b) * UFW + c) * CQ];
                            acc.x = fmaf(w, gv.x, acc.x);
                            acc.y = fmaf(w, gv.y, acc.y);
                            acc.z = fmaf(w, gv.z, acc.z);
                            acc.w = fmaf(w, gv.w, acc.w);
                        }
                    }
                stf4(dx + ((((int64_t)n * g.di + id) * g.hi + ih) * g.wi + iw) * g.x_ld + c0 + 4 * q, acc);
            }
        }
    }
}

// ------------------------------------------------------------------ trilinear x2 backward, marching along D
// The tile kernel above re-reads its fine halo in all three axes ((6*6*34)/(4*4*32) = 2.4x for the 2x2x16 tile; PMC:
// 2.9 GB fetched for the 1.26 GB gradient of the 32-channel level).  The transposed interpolation is separable, so this
// kernel walks a 4x16 coarse column along D: one fine PLANE tile (10 x 34 voxels x 16 channels, fp32) at a time goes
// through LDS (double-buffered, the next plane's pieces are in flight while this one is reduced), every lane folds the 4x4
// in-plane taps of its (coarse h, w, channel quad) into one value S_f, and the D taps are a running sum in registers:
//   dx[i] = wd_i[0] S_{2i-1} + wd_i[1] S_{2i} + wd_i[2] S_{2i+1} + wd_i[3] S_{2i+2}
// so a fine plane is read once per column segment (halo only in H and W: 1.33x, plus 2 planes per MD-plane segment).
constexpr int MTH = 4, MTW = 16, MCQ = 4, MD = 8;            // coarse tile h x w, channel quads per pass, coarse planes per item
constexpr int MFH = 2 * MTH + 2, MFW = 2 * MTW + 2, MPV = MFH * MFW * MCQ;   // 1360 16-byte pieces per fine plane
constexpr int MNP = (MPV + 255) / 256;                       // 6 pieces per lane

template <typename T>
__global__ void __launch_bounds__(256)
upsample2x_bwd_march_kernel(Mri3dUpGeom g, const T* __restrict__ dy, T* __restrict__ dx, int segsD, int tilesH, int tilesW,
                            int cpasses, int nitems) {
    __shared__ float4 pbuf[2][MPV];   // 2 x 21.25 KB
    const int tid = threadIdx.x;
    const int q = tid % MCQ, v = tid / MCQ;
    const int iwl = v % MTW, ihl = v / MTW;
    // blockIdx -> item: consecutive workgroups land on different XCDs, so each XCD gets one contiguous range of items
    // (neighbouring tiles share their H/W halo through that XCD's L2)
    const int per_xcd = (nitems + 7) / 8;
    for (int b = blockIdx.x; b < per_xcd * 8; b += gridDim.x) {
        const int item = (b % 8) * per_xcd + b / 8;
        if (item >= nitems) continue;   // uniform per workgroup
        // channel pass fastest: the passes of one tile read the two halves of the same 128-byte lines, so they run side by
        // side on one XCD and the second reader hits its L2 (32-channel level, 2 x 80x96x80 coarse: 0.60 ms with the passes 120 items apart, 0.37 ms side by side; tile kernel 0.94 ms)
        int t = item;
        const int c0 = (t % cpasses) * (4 * MCQ);
        t /= cpasses;
        const int w0 = (t % tilesW) * MTW;
        t /= tilesW;
        const int h0 = (t % tilesH) * MTH;
        t /= tilesH;
        const int d0 = (t % segsD) * MD;
        const int n = t / segsD;
        const int d1 = min(d0 + MD, g.di);   // coarse planes [d0, d1)
        const T* dn = dy + (int64_t)n * g.dout * g.ho * g.wo * g.y_ld + c0;
        const int64_t plane = (int64_t)g.ho * g.wo * g.y_ld;
        const int ih = h0 + ihl, iw = w0 + iwl;
        const bool vok = ih < g.hi && iw < g.wi && c0 + 4 * q < g.c;
        float wh[4], ww[4];
        up2_axis_weights(ih, g.hi, wh);
        up2_axis_weights(iw, g.wi, ww);
        // this lane's pieces of a plane tile: element offset inside the plane (clamped, so the load is always legal) + mask
        int poff[MNP];
        unsigned pok = 0;
#pragma unroll
        for (int j = 0; j < MNP; ++j) {
            const int idx = j * 256 + tid;
            const int qq = idx % MCQ, fv = idx / MCQ;
            const int fw = fv % MFW, fh = fv / MFW;
            const int oh = 2 * h0 - 1 + fh, ow = 2 * w0 - 1 + fw;
            const bool ok = idx < MPV && (unsigned)oh < (unsigned)g.ho && (unsigned)ow < (unsigned)g.wo && c0 + 4 * qq < g.c;
            pok |= ok ? (1u << j) : 0u;
            const int ch = min(max(oh, 0), g.ho - 1), cw = min(max(ow, 0), g.wo - 1);
            poff[j] = (ch * g.wo + cw) * g.y_ld + (c0 + 4 * qq < g.c ? 4 * qq : 0);
        }
        const int f0 = max(2 * d0 - 1, 0), f1 = min(2 * d1, g.dout - 1);   // fine planes [f0, f1] feed coarse [d0, d1)
        float4 pv[MNP];
        auto fetch = [&](int f) {
            const T* pl = dn + (int64_t)f * plane;
#pragma unroll
            for (int j = 0; j < MNP; ++j) pv[j] = ldf4(pl + poff[j]);
        };
        auto stash = [&](int buf) {
#pragma unroll
            for (int j = 0; j < MNP; ++j) {
                const int idx = j * 256 + tid;
                const bool ok = (pok >> j) & 1u;
                float4 val;
                val.x = ok ? pv[j].x : 0.f; val.y = ok ? pv[j].y : 0.f; val.z = ok ? pv[j].z : 0.f; val.w = ok ? pv[j].w : 0.f;
                if (j < MNP - 1 || idx < MPV) pbuf[buf][idx] = val;
            }
        };
        __syncthreads();   // the previous item's last plane is no longer being read
        fetch(f0);
        stash(f0 & 1);
        __syncthreads();
        float4 accA = make_float4(0.f, 0.f, 0.f, 0.f), accB = make_float4(0.f, 0.f, 0.f, 0.f);
        T* dxo = dx + (((int64_t)n * g.di * g.hi + ih) * g.wi + iw) * g.x_ld + c0 + 4 * q;
        const int64_t xplane = (int64_t)g.hi * g.wi * g.x_ld;
        // (A second register set — plane f+2 in flight while plane f+1 waits for its LDS write, the step barrier ordering LDS only —
        // measured SLOWER: 0.369 -> 0.393 ms fp32, 0.225 -> 0.31 ms bf16 on the 32-channel level; the step is not one exposed round trip.
        // So did the order turned around — lanes own fine positions and apply the D taps to whole-line loads in registers, one LDS
        // pass and one 4 x 4 fold per COARSE plane, 32 channels per pass: parity-green, 0.353 ms fp32 and 0.615 ms bf16 at 256
        // registers per lane.)
        for (int f = f0; f <= f1; ++f) {
            if (f < f1) fetch(f + 1);
            // in-plane 4x4 taps of this lane's coarse (h, w): rows 2*ihl .. 2*ihl+3, columns 2*iwl .. 2*iwl+3 of the tile
            const float4* base = pbuf[f & 1] + ((2 * ihl) * MFW + 2 * iwl) * MCQ + q;
            float4 sf = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int bb = 0; bb < 4; ++bb) {
                float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) {
                    const float4 gv = base[(bb * MFW + cc) * MCQ];
                    r.x = fmaf(ww[cc], gv.x, r.x);
                    r.y = fmaf(ww[cc], gv.y, r.y);
                    r.z = fmaf(ww[cc], gv.z, r.z);
                    r.w = fmaf(ww[cc], gv.w, r.w);
                }
                sf.x = fmaf(wh[bb], r.x, sf.x);
                sf.y = fmaf(wh[bb], r.y, sf.y);
                sf.z = fmaf(wh[bb], r.z, sf.z);
                sf.w = fmaf(wh[bb], r.w, sf.w);
            }
            // D taps (uniform per workgroup): plane f is tap 1/2 of coarse i = f/2 and tap 0 of i+1 (f odd) / tap 3 of i-1 (f even)
            const int i = f >> 1;
            float wdi[4];
            up2_axis_weights(i, g.di, wdi);
            if (f & 1) {
                if (i >= d0) { accA.x = fmaf(wdi[2], sf.x, accA.x); accA.y = fmaf(wdi[2], sf.y, accA.y);
                               accA.z = fmaf(wdi[2], sf.z, accA.z); accA.w = fmaf(wdi[2], sf.w, accA.w); }
                float wn[4];
                up2_axis_weights(i + 1, g.di, wn);
                if (i >= d0) { accB.x = wn[0] * sf.x; accB.y = wn[0] * sf.y; accB.z = wn[0] * sf.z; accB.w = wn[0] * sf.w; }
                else { accA.x = wn[0] * sf.x; accA.y = wn[0] * sf.y; accA.z = wn[0] * sf.z; accA.w = wn[0] * sf.w; }   // f = 2*d0 - 1
            } else {
                if (i > d0) {
                    // tap 3 closes coarse plane i-1
                    float wp[4];
                    up2_axis_weights(i - 1, g.di, wp);
                    accA.x = fmaf(wp[3], sf.x, accA.x); accA.y = fmaf(wp[3], sf.y, accA.y);
                    accA.z = fmaf(wp[3], sf.z, accA.z); accA.w = fmaf(wp[3], sf.w, accA.w);
                    if (vok) stf4(dxo + (int64_t)(i - 1) * xplane, accA);
                    accA = accB;
                }
                if (i < d1) { accA.x = fmaf(wdi[1], sf.x, accA.x); accA.y = fmaf(wdi[1], sf.y, accA.y);
                              accA.z = fmaf(wdi[1], sf.z, accA.z); accA.w = fmaf(wdi[1], sf.w, accA.w); }
            }
            if (f < f1) stash((f + 1) & 1);
            __syncthreads();
        }
        // the last coarse plane of the volume has no tap 3 (its fine plane 2*di does not exist): close it here
        if (d1 == g.di && vok) stf4(dxo + (int64_t)(d1 - 1) * xplane, accA);
    }
}

// ------------------------------------------------------------------ trilinear x2 forward, marching along D
// Fine index 2i+e takes 0.75 of coarse i and 0.25 of coarse i-1 (e = 0) or i+1 (e = 1).  Round 2's tile kernel (one lane = one
// coarse voxel x channel quad writing its 2x2x2 fine voxels from a 3 x 6 x 10 clamped LDS halo) staged 180 halo voxels for the 32
// it owned, waited, then stored — nothing overlapped, and a store instruction's lanes wrote every other fine voxel: 3.6 TB/s
// fp32, 2.8 TB/s bf16 on the 32-channel level (0.383 / 0.25 ms; this kernel 0.283 / 0.156 ms).  The interpolation is separable: with P_c(oh, ow) = the in-plane (h, w) interpolation of coarse plane c at
// a FINE position, fine plane 2c = 0.75 P_c + 0.25 P_{c-1} and fine plane 2c+1 = 0.75 P_c + 0.25 P_{c+1} (ATen's clamped source
// index at the borders = a clamped coarse coordinate: 0.75 a + 0.25 a).  So a workgroup owns a 4 x 16 coarse column = 8 x 32
// fine positions and marches along D: a lane owns FINE positions (consecutive lanes = consecutive 16-byte pieces of consecutive
// fine voxels: a wave-store writes 1 KiB of one fine row contiguously), keeps P_{c-1} in registers, and per coarse plane reads
// four neighbours from a clamped halo plane in LDS (6 x 18 voxels, fp32, double-buffered: the next plane's pieces are in
// flight while this one is interpolated and stored) and stores two fine planes.  The source is read 1.7x (it is an eighth of
// the destination), the destination written once in whole lines.
constexpr int UTH = 4, UTW = 16;                          // coarse column (h, w)
constexpr int UHH = UTH + 2, UHW = UTW + 2, UHV = UHH * UHW;   // clamped halo plane: 108 voxels
constexpr int UMD = 16;                                   // coarse planes per item at most (8 / 4 for small volumes: >= 1024 items)

template <typename T, int VEC, int NPV>   // VEC channels per lane item, NPV items per voxel and channel pass
__global__ void __launch_bounds__(256)
upsample2x_fwd_march_kernel(Mri3dUpGeom g, const T* __restrict__ x, T* __restrict__ y, int segsD, int segl, int tilesH,
                            int tilesW, int cpasses, int nitems) {
    constexpr int PCH = VEC * NPV;                         // channels per pass
    constexpr int NST = (UHV * NPV + 255) / 256;           // staged pieces per lane and plane
    __shared__ __attribute__((aligned(16))) float cbuf[2][UHV * NPV * VEC];
    const int tid = threadIdx.x;
    const int per_xcd = (nitems + 7) / 8;                  // consecutive workgroups land on different XCDs: a contiguous item range each
    for (int b = blockIdx.x; b < per_xcd * 8; b += gridDim.x) {
        const int item = (b % 8) * per_xcd + b / 8;
        if (item >= nitems) continue;   // uniform per workgroup
        int t = item;
        const int c0 = (t % cpasses) * PCH;
        t /= cpasses;
        const int w0 = (t % tilesW) * UTW;
        t /= tilesW;
        const int h0 = (t % tilesH) * UTH;
        t /= tilesH;
        const int d0 = (t % segsD) * segl;
        const int n = t / segsD;
        const int d1 = min(d0 + segl, g.di);   // coarse planes [d0, d1)
        const T* xn = x + (int64_t)n * g.di * g.hi * g.wi * g.x_ld + c0;
        const int xplane = g.hi * g.wi * g.x_ld;
        // the lane's staged pieces of a halo plane: element offset inside the plane (clamped coordinates: always legal)
        int soff[NST];
#pragma unroll
        for (int j = 0; j < NST; ++j) {
            const int idx = min(j * 256 + tid, UHV * NPV - 1);
            const int pc = idx % NPV, hv = idx / NPV;
            const int fw = hv % UHW, fh = hv / UHW;
            const int ch = min(max(h0 - 1 + fh, 0), g.hi - 1), cw = min(max(w0 - 1 + fw, 0), g.wi - 1);
            soff[j] = (ch * g.wi + cw) * g.x_ld + (c0 + pc * VEC < g.c ? pc * VEC : 0);
        }
        V<VEC> sv[NST];
        auto fetch = [&](int c) {
            const T* pl = xn + (int64_t)min(max(c, 0), g.di - 1) * xplane;
#pragma unroll
            for (int j = 0; j < NST; ++j) sv[j].load(pl + soff[j]);
        };
        auto stash = [&](int buf) {
#pragma unroll
            for (int j = 0; j < NST; ++j) {
                const int idx = j * 256 + tid;
                if (j < NST - 1 || idx < UHV * NPV) {
#pragma unroll
                    for (int e = 0; e < VEC; e += 4)
                        *reinterpret_cast<float4*>(&cbuf[buf][idx * VEC + e]) = make_float4(sv[j].v[e], sv[j].v[e + 1], sv[j].v[e + 2], sv[j].v[e + 3]);
                }
            }
        };
        // the lane's items: fine position (fhl, fwl) of the 8 x 32 tile x piece; its four coarse neighbours in the halo plane
        int ia[NPV], ib[NPV], ic[NPV], id[NPV], yoff[NPV];
        unsigned okm = 0;
#pragma unroll
        for (int k = 0; k < NPV; ++k) {
            const int it = k * 256 + tid;
            const int pc = it % NPV, pos = it / NPV;
            const int fwl = pos % (2 * UTW), fhl = pos / (2 * UTW);
            const int hw0 = (fwl >> 1) + 1, hw1 = hw0 + ((fwl & 1) ? 1 : -1), hh0 = (fhl >> 1) + 1, hh1 = hh0 + ((fhl & 1) ? 1 : -1);
            ia[k] = ((hh0 * UHW + hw0) * NPV + pc) * VEC;   // 0.75 x 0.75
            ib[k] = ((hh0 * UHW + hw1) * NPV + pc) * VEC;   // 0.75 x 0.25
            ic[k] = ((hh1 * UHW + hw0) * NPV + pc) * VEC;   // 0.25 x 0.75
            id[k] = ((hh1 * UHW + hw1) * NPV + pc) * VEC;   // 0.25 x 0.25
            const int oh = 2 * h0 + fhl, ow = 2 * w0 + fwl;
            const bool ok = oh < g.ho && ow < g.wo && c0 + pc * VEC < g.c;
            okm |= ok ? (1u << k) : 0u;
            yoff[k] = ok ? (oh * g.wo + ow) * g.y_ld + c0 + pc * VEC : 0;
        }
        T* yn = y + (int64_t)n * g.dout * g.ho * g.wo * g.y_ld;
        const int64_t yplane = (int64_t)g.ho * g.wo * g.y_ld;

        __syncthreads();   // the previous item's last plane is no longer being read
        fetch(d0 - 1);
        stash(0);
        __syncthreads();
        V<VEC> pp[NPV];    // P of the previous coarse plane
        for (int c = d0 - 1; c <= d1; ++c) {
            const int buf = (c - (d0 - 1)) & 1;
            if (c < d1) fetch(c + 1);
#pragma unroll
            for (int k = 0; k < NPV; ++k) {
                V<VEC> pc_;
#pragma unroll
                for (int e = 0; e < VEC; e += 4) {
                    const float4 a = *reinterpret_cast<const float4*>(&cbuf[buf][ia[k] + e]);
                    const float4 bq = *reinterpret_cast<const float4*>(&cbuf[buf][ib[k] + e]);
                    const float4 cq = *reinterpret_cast<const float4*>(&cbuf[buf][ic[k] + e]);
                    const float4 dq = *reinterpret_cast<const float4*>(&cbuf[buf][id[k] + e]);
                    pc_.v[e + 0] = 0.75f * fmaf(0.75f, a.x, 0.25f * bq.x) + 0.25f * fmaf(0.75f, cq.x, 0.25f * dq.x);
                    pc_.v[e + 1] = 0.75f * fmaf(0.75f, a.y, 0.25f * bq.y) + 0.25f * fmaf(0.75f, cq.y, 0.25f * dq.y);
                    pc_.v[e + 2] = 0.75f * fmaf(0.75f, a.z, 0.25f * bq.z) + 0.25f * fmaf(0.75f, cq.z, 0.25f * dq.z);
                    pc_.v[e + 3] = 0.75f * fmaf(0.75f, a.w, 0.25f * bq.w) + 0.25f * fmaf(0.75f, cq.w, 0.25f * dq.w);
                }
                if (c >= d0 && ((okm >> k) & 1u)) {
                    V<VEC> o;
                    if (c > d0) {        // fine plane 2(c-1)+1 = 0.75 P_{c-1} + 0.25 P_c
#pragma unroll
                        for (int e = 0; e < VEC; ++e) o.v[e] = fmaf(0.75f, pp[k].v[e], 0.25f * pc_.v[e]);
                        o.store(yn + (int64_t)(2 * c - 1) * yplane + yoff[k]);
                    }
                    if (c < d1) {        // fine plane 2c = 0.75 P_c + 0.25 P_{c-1}
#pragma unroll
                        for (int e = 0; e < VEC; ++e) o.v[e] = fmaf(0.75f, pc_.v[e], 0.25f * pp[k].v[e]);
                        o.store(yn + (int64_t)(2 * c) * yplane + yoff[k]);
                    }
                }
                pp[k] = pc_;
            }
            if (c < d1) stash(buf ^ 1);
            __syncthreads();
        }
    }
}

static inline bool up2x_fast_ok(const Mri3dUpGeom& g) {
    return g.mode == MRI3D_UP_TRILINEAR && !g.align_corners && g.dout == 2 * g.di && g.ho == 2 * g.hi && g.wo == 2 * g.wi &&
           g.rd == 0.5f && g.rh == 0.5f && g.rw == 0.5f && g.c % 4 == 0 && g.x_ld % 4 == 0 && g.y_ld % 4 == 0;
}

// rows of H per slab so that one slab is ~8 passes of a 256-thread block, and the resulting grid size
static inline void slab_plan(int nd, int h, int w, int cv, int& hch, int& grid) {
    int64_t per_row = (int64_t)w * cv;
    hch = (int)std::max<int64_t>(1, std::min<int64_t>(h, 2048 / std::max<int64_t>(per_row, 1)));
    int64_t slabs = (int64_t)nd * ((h + hch - 1) / hch);
    grid = (int)std::min<int64_t>(slabs, 8192);
}

static inline bool vec_ok(int dtype, int c, int a_ld, int b_ld, const void* a, const void* b) {
    return c % 4 == 0 && a_ld % 4 == 0 && b_ld % 4 == 0 && aligned_vec4(dtype, a, b);
}

}  // namespace mri3d

using namespace mri3d;

static int pool_check(const Mri3dPoolGeom* g, const char* who) {
    MRI3D_REQUIRE(g != nullptr, MRI3D_EINVAL, "%s: null geometry", who);
    MRI3D_REQUIRE(g->dtype == MRI3D_F32 || g->dtype == MRI3D_BF16, MRI3D_ENOTSUP, "%s: unknown dtype %d", who, g->dtype);
    MRI3D_REQUIRE(g->n > 0 && g->c > 0 && g->di > 0 && g->hi > 0 && g->wi > 0 && g->dout > 0 && g->ho > 0 && g->wo > 0,
                  MRI3D_EINVAL, "%s: empty tensor", who);
    MRI3D_REQUIRE(g->kd * g->kh * g->kw <= 256 && g->kd > 0 && g->kh > 0 && g->kw > 0, MRI3D_ENOTSUP,
                  "%s: window volume must be <= 256", who);
    MRI3D_REQUIRE(g->sd > 0 && g->sh > 0 && g->sw > 0 && g->x_ld >= g->c && g->y_ld >= g->c, MRI3D_EINVAL,
                  "%s: bad stride/pitch", who);
    MRI3D_REQUIRE((g->dout - 1) * g->sd - g->pd < g->di && (g->ho - 1) * g->sh - g->ph < g->hi &&
                      (g->wo - 1) * g->sw - g->pw < g->wi && 2 * g->pd <= g->kd && 2 * g->ph <= g->kh && 2 * g->pw <= g->kw,
                  MRI3D_EINVAL, "%s: output dims inconsistent with input", who);
    return MRI3D_OK;
}

extern "C" int mri3d_maxpool3d_fwd(const Mri3dPoolGeom* g, const void* x, void* y, uint8_t* idx,
                                   mri3d_stream_t stream) {
    int rc = pool_check(g, "maxpool3d_fwd");
    if (rc) return rc;
    MRI3D_REQUIRE(x && y && idx, MRI3D_EINVAL, "maxpool3d_fwd: null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    bool v4 = vec_ok(g->dtype, g->c, g->x_ld, g->y_ld, x, y) && (reinterpret_cast<uintptr_t>(idx) & 3) == 0;
    int hch, grid;
    const bool v8 = v4 && g->dtype == MRI3D_BF16 && g->c % 8 == 0 && g->x_ld % 8 == 0 && g->y_ld % 8 == 0 && aligned16(x, y) &&
                    (reinterpret_cast<uintptr_t>(idx) & 7) == 0;
    slab_plan(g->n * g->dout, g->ho, g->wo, g->c / (v8 ? 8 : (v4 ? 4 : 1)), hch, grid);
    if (v4 && pool2_ok(*g, v8 ? 8 : 4)) {   // MaxPool3d(2) on even extents: lanes walk input columns (whole lines per wave-load)
        slab_plan(g->n * g->dout, g->ho, g->wi, g->c / (v8 ? 8 : 4), hch, grid);
        MRI3D_DISPATCH_DTYPE(g->dtype, T, {
            if constexpr (sizeof(T) == 2) {
                if (v8) hipLaunchKernelGGL((maxpool2_fwd_kernel<T, 8>), dim3(grid), dim3(256), 0, s, *g, (const T*)x, (T*)y, idx, hch);
            }
            if (!v8) hipLaunchKernelGGL((maxpool2_fwd_kernel<T, 4>), dim3(grid), dim3(256), 0, s, *g, (const T*)x, (T*)y, idx, hch);
        });
        return check_launch("maxpool3d_fwd");
    }
    MRI3D_DISPATCH_DTYPE(g->dtype, T, {
        if constexpr (sizeof(T) == 2) {
            if (v8) hipLaunchKernelGGL((maxpool_fwd_kernel<T, 8>), dim3(grid), dim3(256), 0, s, *g, (const T*)x, (T*)y, idx, hch);
        }
        if (v8) {
        } else if (v4)
            hipLaunchKernelGGL((maxpool_fwd_kernel<T, 4>), dim3(grid), dim3(256), 0, s, *g, (const T*)x, (T*)y, idx, hch);
        else
            hipLaunchKernelGGL((maxpool_fwd_kernel<T, 1>), dim3(grid), dim3(256), 0, s, *g, (const T*)x, (T*)y, idx, hch);
    });
    return check_launch("maxpool3d_fwd");
}

static int maxpool_bwd_impl(const Mri3dPoolGeom* g, const void* dy, const uint8_t* idx, const void* addend, int32_t a_ld,
                            void* dx, mri3d_stream_t stream) {
    int rc = pool_check(g, "maxpool3d_bwd");
    if (rc) return rc;
    MRI3D_REQUIRE(dy && dx && idx, MRI3D_EINVAL, "maxpool3d_bwd: null pointer");
    MRI3D_REQUIRE(addend == nullptr || a_ld >= g->c, MRI3D_EINVAL, "maxpool3d_bwd: addend pitch %d < %d channels", a_ld, g->c);
    hipStream_t s = static_cast<hipStream_t>(stream);
    bool v4 = vec_ok(g->dtype, g->c, g->x_ld, g->y_ld, dx, dy) && (reinterpret_cast<uintptr_t>(idx) & 3) == 0 &&
              (addend == nullptr || (a_ld % 4 == 0 && aligned_vec4(g->dtype, addend)));
    int hch, grid;
    const bool v8 = v4 && g->dtype == MRI3D_BF16 && g->c % 8 == 0 && g->x_ld % 8 == 0 && g->y_ld % 8 == 0 && aligned16(dx, dy) &&
                    (reinterpret_cast<uintptr_t>(idx) & 7) == 0 && (addend == nullptr || (a_ld % 8 == 0 && aligned16(addend)));
    slab_plan(g->n * g->di, g->hi, g->wi, g->c / (v8 ? 8 : (v4 ? 4 : 1)), hch, grid);
    MRI3D_DISPATCH_DTYPE(g->dtype, T, {
        if constexpr (sizeof(T) == 2) {
            if (v8) hipLaunchKernelGGL((maxpool_bwd_kernel<T, 8>), dim3(grid), dim3(256), 0, s, *g, (const T*)dy, idx, (T*)dx, hch,
                                       (const T*)addend, a_ld);
        }
        if (v8) {
        } else if (v4)
            hipLaunchKernelGGL((maxpool_bwd_kernel<T, 4>), dim3(grid), dim3(256), 0, s, *g, (const T*)dy, idx, (T*)dx, hch,
                               (const T*)addend, a_ld);
        else
            hipLaunchKernelGGL((maxpool_bwd_kernel<T, 1>), dim3(grid), dim3(256), 0, s, *g, (const T*)dy, idx, (T*)dx, hch,
                               (const T*)addend, a_ld);
    });
    return check_launch("maxpool3d_bwd");
}

extern "C" int mri3d_maxpool3d_bwd(const Mri3dPoolGeom* g, const void* dy, const uint8_t* idx, void* dx,
                                   mri3d_stream_t stream) {
    return maxpool_bwd_impl(g, dy, idx, nullptr, 0, dx, stream);
}

extern "C" int mri3d_maxpool3d_bwd_add(const Mri3dPoolGeom* g, const void* dy, const uint8_t* idx, const void* addend,
                                       int32_t addend_ld, void* dx, mri3d_stream_t stream) {
    MRI3D_REQUIRE(addend != nullptr, MRI3D_EINVAL, "maxpool3d_bwd_add: null addend");
    return maxpool_bwd_impl(g, dy, idx, addend, addend_ld, dx, stream);
}

static int up_check(const Mri3dUpGeom* g, const char* who) {
    MRI3D_REQUIRE(g != nullptr, MRI3D_EINVAL, "%s: null geometry", who);
    MRI3D_REQUIRE(g->dtype == MRI3D_F32 || g->dtype == MRI3D_BF16, MRI3D_ENOTSUP, "%s: unknown dtype %d", who, g->dtype);
    MRI3D_REQUIRE(g->n > 0 && g->c > 0 && g->di > 0 && g->hi > 0 && g->wi > 0 && g->dout > 0 && g->ho > 0 && g->wo > 0,
                  MRI3D_EINVAL, "%s: empty tensor", who);
    MRI3D_REQUIRE(g->mode == MRI3D_UP_NEAREST || g->mode == MRI3D_UP_TRILINEAR, MRI3D_EINVAL, "%s: bad mode %d", who,
                  g->mode);
    MRI3D_REQUIRE(g->x_ld >= g->c && g->y_ld >= g->c, MRI3D_EINVAL, "%s: bad pitch", who);
    MRI3D_REQUIRE(g->rd >= 0.f && g->rh >= 0.f && g->rw >= 0.f, MRI3D_EINVAL, "%s: negative scale", who);
    return MRI3D_OK;
}

extern "C" size_t mri3d_upsample3d_workspace_bytes(const Mri3dUpGeom* g) {
    if (!g) return 0;
    return (size_t)2 * (g->di + g->hi + g->wi) * sizeof(int);
}

extern "C" int mri3d_upsample3d_fwd(const Mri3dUpGeom* g, const void* x, void* y, mri3d_stream_t stream) {
    int rc = up_check(g, "upsample3d_fwd");
    if (rc) return rc;
    MRI3D_REQUIRE(x && y, MRI3D_EINVAL, "upsample3d_fwd: null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    static const int no_fast_f = tuning_knob("MRI3D_UP_GENERIC", 0);   // tuning aid (A/B)
    if (!no_fast_f && up2x_fast_ok(*g) && aligned_vec4(g->dtype, x, y)) {
        // marching kernel: lane items of 16 bytes where the tensors allow it (bf16: channels and pitches in whole octets)
        const bool oct = g->dtype == MRI3D_BF16 && g->c % 8 == 0 && g->x_ld % 8 == 0 && g->y_ld % 8 == 0 &&
                         ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0;
        const int vec = oct ? 8 : 4, pieces = g->c / vec;
        const int npv = pieces >= 8 ? 8 : pieces >= 4 ? 4 : pieces >= 2 ? 2 : 1;
        const int mtilesH = cdiv(g->hi, UTH), mtilesW = cdiv(g->wi, UTW), mpasses = cdiv(pieces, npv);
        int segl = UMD;   // every segment stages two planes more than it owns: long segments unless the grid would not fill the chip
        while (segl > 4 && (int64_t)g->n * cdiv(g->di, segl) * mpasses * mtilesH * mtilesW < 1024) segl /= 2;
        const int segsD = cdiv(g->di, segl);
        const int64_t items = (int64_t)g->n * segsD * mpasses * mtilesH * mtilesW;
        if (items <= 0x7ffffff0 && (int64_t)g->ho * g->wo * g->y_ld <= 0x7fffffff &&
            (int64_t)g->hi * g->wi * g->x_ld <= 0x7fffffff) {
            const int grid = (int)std::min<int64_t>((items + 7) / 8 * 8, 256 * 4 * 8);
#define MRI3D_UPF(Tv, VECv, NPVv)                                                                                      \
    hipLaunchKernelGGL((upsample2x_fwd_march_kernel<Tv, VECv, NPVv>), dim3(grid), dim3(256), 0, s, *g, (const Tv*)x,  \
                       (Tv*)y, segsD, segl, mtilesH, mtilesW, mpasses, (int)items)
#define MRI3D_UPF_N(Tv, VECv)                                                                                          \
    do {                                                                                                              \
        if (npv == 8) MRI3D_UPF(Tv, VECv, 8);                                                                         \
        else if (npv == 4) MRI3D_UPF(Tv, VECv, 4);                                                                    \
        else if (npv == 2) MRI3D_UPF(Tv, VECv, 2);                                                                    \
        else MRI3D_UPF(Tv, VECv, 1);                                                                                  \
    } while (0)
            if (g->dtype == MRI3D_F32) MRI3D_UPF_N(float, 4);
            else if (oct) MRI3D_UPF_N(bf16_t, 8);
            else MRI3D_UPF_N(bf16_t, 4);
#undef MRI3D_UPF_N
#undef MRI3D_UPF
            return check_launch("upsample3d_fwd(2x march)");
        }
    }
    bool v4 = vec_ok(g->dtype, g->c, g->x_ld, g->y_ld, x, y);
    int hch, grid;
    slab_plan(g->n * g->dout, g->ho, g->wo, g->c / (v4 ? 4 : 1), hch, grid);
    MRI3D_DISPATCH_DTYPE(g->dtype, T, {
        if (v4)
            hipLaunchKernelGGL((upsample_fwd_kernel<T, 4>), dim3(grid), dim3(256), 0, s, *g, (const T*)x, (T*)y, hch);
        else
            hipLaunchKernelGGL((upsample_fwd_kernel<T, 1>), dim3(grid), dim3(256), 0, s, *g, (const T*)x, (T*)y, hch);
    });
    return check_launch("upsample3d_fwd");
}

extern "C" int mri3d_upsample3d_bwd(const Mri3dUpGeom* g, const void* dy, void* dx, void* workspace, size_t ws_bytes,
                                    mri3d_stream_t stream) {
    int rc = up_check(g, "upsample3d_bwd");
    if (rc) return rc;
    MRI3D_REQUIRE(dy && dx, MRI3D_EINVAL, "upsample3d_bwd: null pointer");
    MRI3D_REQUIRE(workspace && ws_bytes >= mri3d_upsample3d_workspace_bytes(g), MRI3D_EWORKSPACE,
                  "upsample3d_bwd: workspace %zu < %zu", ws_bytes, mri3d_upsample3d_workspace_bytes(g));
    hipStream_t s = static_cast<hipStream_t>(stream);
    static const int no_fast = tuning_knob("MRI3D_UP_GENERIC", 0);   // tuning aid (A/B)
    if (!no_fast && up2x_fast_ok(*g) && aligned_vec4(g->dtype, dx, dy)) {
        static const int march = tuning_knob("MRI3D_UP_MARCH", 1);   // tuning aid (A/B)
        if (march && g->c % 16 == 0) {
            const int segsD = cdiv(g->di, MD), tilesH = cdiv(g->hi, MTH), tilesW = cdiv(g->wi, MTW), cpasses = g->c / 16;
            const int64_t items = (int64_t)g->n * segsD * cpasses * tilesH * tilesW;
            if (items <= 0x7ffffff0 && (int64_t)g->ho * g->wo * g->y_ld <= 0x7fffffff) {
                const int grid = (int)std::min<int64_t>((items + 7) / 8 * 8, 256 * 3 * 8);
                MRI3D_DISPATCH_DTYPE(g->dtype, T, {
                    hipLaunchKernelGGL(upsample2x_bwd_march_kernel<T>, dim3(grid), dim3(256), 0, s, *g, (const T*)dy, (T*)dx,
                                       segsD, tilesH, tilesW, cpasses, (int)items);
                });
                return check_launch("upsample3d_bwd(2x march)");
            }
        }
        // 16-channel passes pay off for fp32 (0.93 vs 1.25 ms on the c32 level); bf16 already reads 32-byte slices with 8
        // channels... measured 0.68 ms (CQ = 2) vs 0.96 ms (CQ = 4), so it keeps the 8-channel tile
        const bool wide = g->c % 16 == 0 && g->dtype == MRI3D_F32;
        const int utd = wide ? Up2Tile<4>::TD : Up2Tile<2>::TD, uth = wide ? Up2Tile<4>::TH : Up2Tile<2>::TH, utw = 16;
        const int tilesD = cdiv(g->di, utd), tilesH = cdiv(g->hi, uth), tilesW = cdiv(g->wi, utw);
        const int64_t nt = (int64_t)g->n * tilesD * tilesH * tilesW;
        if (nt <= 0x7fffffff) {
            const size_t smem = (size_t)(2 * utd + 2) * (2 * uth + 2) * (2 * utw + 2) * (wide ? 4 : 2) * sizeof(float4);
            const int grid = (int)std::min<int64_t>(nt, 2048);
#define MRI3D_UP2B(CQv)                                                                                               \
    {                                                                                                                 \
        auto kern = upsample2x_bwd_kernel<T, CQv>;                                                                    \
        static const hipError_t attr_ = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                      \
                                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);   \
        (void)attr_;   /* once per kernel (smem is a constant of the instantiation), not per launch */                \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), smem, s, *g, (const T*)dy, (T*)dx, tilesD, tilesH, tilesW,    \
                           (int)nt);                                                                                  \
    }
            MRI3D_DISPATCH_DTYPE(g->dtype, T, {
                if (wide) MRI3D_UP2B(4) else MRI3D_UP2B(2)
            });
#undef MRI3D_UP2B
            return check_launch("upsample3d_bwd(2x)");
        }
    }
    bool v4 = vec_ok(g->dtype, g->c, g->x_ld, g->y_ld, dx, dy);
    int hch, grid;
    slab_plan(g->n * g->di, g->hi, g->wi, g->c / (v4 ? 4 : 1), hch, grid);
    // nearest neighbour with output = S x input in every axis (S = 2, 4): the plain S^3 box sum
    const int sc = g->di > 0 ? g->dout / g->di : 0;
    if (!no_fast && g->mode == MRI3D_UP_NEAREST && v4 && (sc == 2 || sc == 4) && g->dout == sc * g->di && g->ho == sc * g->hi &&
        g->wo == sc * g->wi) {
        MRI3D_DISPATCH_DTYPE(g->dtype, T, {
            if (sc == 4)
                hipLaunchKernelGGL((upsample_nearest_int_bwd_kernel<T, 4>), dim3(grid), dim3(256), 0, s, *g, (const T*)dy, (T*)dx, hch);
            else
                hipLaunchKernelGGL((upsample_nearest_int_bwd_kernel<T, 2>), dim3(grid), dim3(256), 0, s, *g, (const T*)dy, (T*)dx, hch);
        });
        return check_launch("upsample3d_bwd(nearest, integer scale)");
    }
    int* tab = static_cast<int*>(workspace);
    hipLaunchKernelGGL(upsample_tables_kernel, dim3(3), dim3(256), 0, s, *g, tab);
    MRI3D_DISPATCH_DTYPE(g->dtype, T, {
        if (v4)
            hipLaunchKernelGGL((upsample_bwd_kernel<T, 4>), dim3(grid), dim3(256), 0, s, *g, (const T*)dy, (T*)dx, tab, hch);
        else
            hipLaunchKernelGGL((upsample_bwd_kernel<T, 1>), dim3(grid), dim3(256), 0, s, *g, (const T*)dy, (T*)dx, tab, hch);
    });
    return check_launch("upsample3d_bwd");
}
