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

namespace mri3d {

template <int VEC>
struct V {
    float v[VEC];
    __device__ __forceinline__ void load(const float* p) {
        if (VEC == 4) {
            float4 t = *reinterpret_cast<const float4*>(p);
            v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
        } else {
            v[0] = *p;
        }
    }
    __device__ __forceinline__ void store(float* p) const {
        if (VEC == 4) *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
        else *p = v[0];
    }
};

// ------------------------------------------------------------------ max pool forward
template <int VEC>
__global__ void __launch_bounds__(256)
maxpool_fwd_kernel(Mri3dPoolGeom g, const float* __restrict__ x, float* __restrict__ y, uint8_t* __restrict__ idx) {
    const int CV = g.c / VEC;
    const int64_t total = (int64_t)g.n * g.dout * g.ho * g.wo * CV;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int cv = (int)(i % CV);
        int64_t ov = i / CV;
        int ow = (int)(ov % g.wo);
        int64_t t = ov / g.wo;
        int oh = (int)(t % g.ho);
        t /= g.ho;
        int od = (int)(t % g.dout);
        int n = (int)(t / g.dout);
        float best[VEC];
        int bi[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) { best[j] = -INFINITY; bi[j] = 0; }
        bool first = true;
        for (int kd = 0; kd < g.kd; ++kd) {
            int id = od * g.sd - g.pd + kd;
            if ((unsigned)id >= (unsigned)g.di) continue;
            for (int kh = 0; kh < g.kh; ++kh) {
                int ih = oh * g.sh - g.ph + kh;
                if ((unsigned)ih >= (unsigned)g.hi) continue;
                for (int kw = 0; kw < g.kw; ++kw) {
                    int iw = ow * g.sw - g.pw + kw;
                    if ((unsigned)iw >= (unsigned)g.wi) continue;
                    V<VEC> xv;
                    xv.load(x + ((((int64_t)n * g.di + id) * g.hi + ih) * g.wi + iw) * g.x_ld + cv * VEC);
                    int tap = (kd * g.kh + kh) * g.kw + kw;
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
        o.store(y + ov * g.y_ld + cv * VEC);
        uint8_t* ip = idx + ov * g.c + cv * VEC;
        if (VEC == 4) {
            *reinterpret_cast<uint32_t*>(ip) = (uint32_t)bi[0] | ((uint32_t)bi[1] << 8) | ((uint32_t)bi[VEC > 2 ? 2 : 0] << 16) |
                                               ((uint32_t)bi[VEC > 3 ? 3 : 0] << 24);
        } else {
            ip[0] = (uint8_t)bi[0];
        }
    }
}

// ------------------------------------------------------------------ max pool backward (gather over covering windows)
template <int VEC>
__global__ void __launch_bounds__(256)
maxpool_bwd_kernel(Mri3dPoolGeom g, const float* __restrict__ dy, const uint8_t* __restrict__ idx,
                   float* __restrict__ dx) {
    const int CV = g.c / VEC;
    const int64_t total = (int64_t)g.n * g.di * g.hi * g.wi * CV;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int cv = (int)(i % CV);
        int64_t iv = i / CV;
        int iw = (int)(iv % g.wi);
        int64_t t = iv / g.wi;
        int ih = (int)(t % g.hi);
        t /= g.hi;
        int id = (int)(t % g.di);
        int n = (int)(t / g.di);
        float acc[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
        // windows od with od*sd - pd <= id <= od*sd - pd + kd - 1
        int od_lo = id + g.pd - g.kd + 1; od_lo = od_lo <= 0 ? 0 : (od_lo + g.sd - 1) / g.sd;
        int od_hi = (id + g.pd) / g.sd; if (od_hi > g.dout - 1) od_hi = g.dout - 1;
        int oh_lo = ih + g.ph - g.kh + 1; oh_lo = oh_lo <= 0 ? 0 : (oh_lo + g.sh - 1) / g.sh;
        int oh_hi = (ih + g.ph) / g.sh; if (oh_hi > g.ho - 1) oh_hi = g.ho - 1;
        int ow_lo = iw + g.pw - g.kw + 1; ow_lo = ow_lo <= 0 ? 0 : (ow_lo + g.sw - 1) / g.sw;
        int ow_hi = (iw + g.pw) / g.sw; if (ow_hi > g.wo - 1) ow_hi = g.wo - 1;
        for (int od = od_lo; od <= od_hi; ++od) {
            int kd = id - (od * g.sd - g.pd);
            for (int oh = oh_lo; oh <= oh_hi; ++oh) {
                int kh = ih - (oh * g.sh - g.ph);
                for (int ow = ow_lo; ow <= ow_hi; ++ow) {
                    int kw = iw - (ow * g.sw - g.pw);
                    int tap = (kd * g.kh + kh) * g.kw + kw;
                    int64_t ov = (((int64_t)n * g.dout + od) * g.ho + oh) * g.wo + ow;
                    V<VEC> gv;
                    gv.load(dy + ov * g.y_ld + cv * VEC);
                    const uint8_t* ip = idx + ov * g.c + cv * VEC;
                    if (VEC == 4) {
                        uint32_t pk = *reinterpret_cast<const uint32_t*>(ip);
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
        o.store(dx + iv * g.x_ld + cv * VEC);
    }
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

template <int VEC>
__global__ void __launch_bounds__(256)
upsample_fwd_kernel(Mri3dUpGeom g, const float* __restrict__ x, float* __restrict__ y) {
    const int CV = g.c / VEC;
    const int64_t total = (int64_t)g.n * g.dout * g.ho * g.wo * CV;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int cv = (int)(i % CV);
        int64_t ov = i / CV;
        int ow = (int)(ov % g.wo);
        int64_t t = ov / g.wo;
        int oh = (int)(t % g.ho);
        t /= g.ho;
        int od = (int)(t % g.dout);
        int n = (int)(t / g.dout);
        const float* xn = x + (int64_t)n * g.di * g.hi * g.wi * g.x_ld + cv * VEC;
        V<VEC> o;
        if (g.mode == MRI3D_UP_NEAREST) {
            int id = near_src(od, g.rd, g.di), ih = near_src(oh, g.rh, g.hi), iw = near_src(ow, g.rw, g.wi);
            o.load(xn + (((int64_t)id * g.hi + ih) * g.wi + iw) * g.x_ld);
        } else {
            Lin Ld_ = lin_src(od, g.rd, g.di, g.align_corners);
            Lin Lh = lin_src(oh, g.rh, g.hi, g.align_corners);
            Lin Lw = lin_src(ow, g.rw, g.wi, g.align_corners);
#pragma unroll
            for (int j = 0; j < VEC; ++j) o.v[j] = 0.f;
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                int id = a ? Ld_.i1 : Ld_.i0;
                float wd = a ? Ld_.l1 : Ld_.l0;
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    int ih = b ? Lh.i1 : Lh.i0;
                    float wh = wd * (b ? Lh.l1 : Lh.l0);
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        int iw = c ? Lw.i1 : Lw.i0;
                        float w = wh * (c ? Lw.l1 : Lw.l0);
                        V<VEC> xv;
                        xv.load(xn + (((int64_t)id * g.hi + ih) * g.wi + iw) * g.x_ld);
#pragma unroll
                        for (int j = 0; j < VEC; ++j) o.v[j] = fmaf(w, xv.v[j], o.v[j]);
                    }
                }
            }
        }
        o.store(y + ov * g.y_ld + cv * VEC);
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

template <int VEC>
__global__ void __launch_bounds__(256)
upsample_bwd_kernel(Mri3dUpGeom g, const float* __restrict__ dy, float* __restrict__ dx, const int* __restrict__ tab) {
    const int CV = g.c / VEC;
    const int* lo_d = tab; const int* hi_d = tab + g.di;
    const int* lo_h = tab + 2 * g.di; const int* hi_h = lo_h + g.hi;
    const int* lo_w = tab + 2 * g.di + 2 * g.hi; const int* hi_w = lo_w + g.wi;
    const int64_t total = (int64_t)g.n * g.di * g.hi * g.wi * CV;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int cv = (int)(i % CV);
        int64_t iv = i / CV;
        int iw = (int)(iv % g.wi);
        int64_t t = iv / g.wi;
        int ih = (int)(t % g.hi);
        t /= g.hi;
        int id = (int)(t % g.di);
        int n = (int)(t / g.di);
        float acc[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
        const float* dn = dy + (int64_t)n * g.dout * g.ho * g.wo * g.y_ld + cv * VEC;
        const int d0 = lo_d[id], d1 = hi_d[id], h0 = lo_h[ih], h1 = hi_h[ih], w0 = lo_w[iw], w1 = hi_w[iw];
        for (int od = d0; od <= d1; ++od) {
            float wd = up_weight(od, id, g.rd, g.di, g.mode, g.align_corners);
            if (wd == 0.f) continue;
            for (int oh = h0; oh <= h1; ++oh) {
                float wh = wd * up_weight(oh, ih, g.rh, g.hi, g.mode, g.align_corners);
                if (wh == 0.f) continue;
                for (int ow = w0; ow <= w1; ++ow) {
                    float w = wh * up_weight(ow, iw, g.rw, g.wi, g.mode, g.align_corners);
                    if (w == 0.f) continue;
                    V<VEC> gv;
                    gv.load(dn + (((int64_t)od * g.ho + oh) * g.wo + ow) * g.y_ld);
#pragma unroll
                    for (int j = 0; j < VEC; ++j) acc[j] = fmaf(w, gv.v[j], acc[j]);
                }
            }
        }
        V<VEC> o;
#pragma unroll
        for (int j = 0; j < VEC; ++j) o.v[j] = acc[j];
        o.store(dx + iv * g.x_ld + cv * VEC);
    }
}

static inline bool vec_ok(int c, int a_ld, int b_ld, const void* a, const void* b) {
    return c % 4 == 0 && a_ld % 4 == 0 && b_ld % 4 == 0 &&
           ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 15) == 0;
}

}  // namespace mri3d

using namespace mri3d;

static int pool_check(const Mri3dPoolGeom* g, const char* who) {
    MRI3D_REQUIRE(g != nullptr, MRI3D_EINVAL, "%s: null geometry", who);
    MRI3D_REQUIRE(g->dtype == MRI3D_F32, MRI3D_ENOTSUP, "%s: only MRI3D_F32 is implemented", who);
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
    bool v4 = vec_ok(g->c, g->x_ld, g->y_ld, x, y) && (reinterpret_cast<uintptr_t>(idx) & 3) == 0;
    int64_t total = (int64_t)g->n * g->dout * g->ho * g->wo * (g->c / (v4 ? 4 : 1));
    int grid = stream_grid(total, 256);
    if (v4)
        hipLaunchKernelGGL(maxpool_fwd_kernel<4>, dim3(grid), dim3(256), 0, s, *g, (const float*)x, (float*)y, idx);
    else
        hipLaunchKernelGGL(maxpool_fwd_kernel<1>, dim3(grid), dim3(256), 0, s, *g, (const float*)x, (float*)y, idx);
    return check_launch("maxpool3d_fwd");
}

extern "C" int mri3d_maxpool3d_bwd(const Mri3dPoolGeom* g, const void* dy, const uint8_t* idx, void* dx,
                                   mri3d_stream_t stream) {
    int rc = pool_check(g, "maxpool3d_bwd");
    if (rc) return rc;
    MRI3D_REQUIRE(dy && dx && idx, MRI3D_EINVAL, "maxpool3d_bwd: null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    bool v4 = vec_ok(g->c, g->x_ld, g->y_ld, dx, dy) && (reinterpret_cast<uintptr_t>(idx) & 3) == 0;
    int64_t total = (int64_t)g->n * g->di * g->hi * g->wi * (g->c / (v4 ? 4 : 1));
    int grid = stream_grid(total, 256);
    if (v4)
        hipLaunchKernelGGL(maxpool_bwd_kernel<4>, dim3(grid), dim3(256), 0, s, *g, (const float*)dy, idx, (float*)dx);
    else
        hipLaunchKernelGGL(maxpool_bwd_kernel<1>, dim3(grid), dim3(256), 0, s, *g, (const float*)dy, idx, (float*)dx);
    return check_launch("maxpool3d_bwd");
}

static int up_check(const Mri3dUpGeom* g, const char* who) {
    MRI3D_REQUIRE(g != nullptr, MRI3D_EINVAL, "%s: null geometry", who);
    MRI3D_REQUIRE(g->dtype == MRI3D_F32, MRI3D_ENOTSUP, "%s: only MRI3D_F32 is implemented", who);
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
    bool v4 = vec_ok(g->c, g->x_ld, g->y_ld, x, y);
    int64_t total = (int64_t)g->n * g->dout * g->ho * g->wo * (g->c / (v4 ? 4 : 1));
    int grid = stream_grid(total, 256);
    if (v4)
        hipLaunchKernelGGL(upsample_fwd_kernel<4>, dim3(grid), dim3(256), 0, s, *g, (const float*)x, (float*)y);
    else
        hipLaunchKernelGGL(upsample_fwd_kernel<1>, dim3(grid), dim3(256), 0, s, *g, (const float*)x, (float*)y);
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
    int* tab = static_cast<int*>(workspace);
    hipLaunchKernelGGL(upsample_tables_kernel, dim3(3), dim3(256), 0, s, *g, tab);
    bool v4 = vec_ok(g->c, g->x_ld, g->y_ld, dx, dy);
    int64_t total = (int64_t)g->n * g->di * g->hi * g->wi * (g->c / (v4 ? 4 : 1));
    int grid = stream_grid(total, 256);
    if (v4)
        hipLaunchKernelGGL(upsample_bwd_kernel<4>, dim3(grid), dim3(256), 0, s, *g, (const float*)dy, (float*)dx, tab);
    else
        hipLaunchKernelGGL(upsample_bwd_kernel<1>, dim3(grid), dim3(256), 0, s, *g, (const float*)dy, (float*)dx, tab);
    return check_launch("upsample3d_bwd");
}
