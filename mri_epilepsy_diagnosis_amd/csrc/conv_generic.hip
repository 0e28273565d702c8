// conv_generic.hip — direct NDHWC Conv3d forward / data-gradient / weight-gradient for ANY kernel size,
// stride, padding and dilation (separable (k,1,1) convs of AE_model.py:9-26, strided convs of
// modified_3dunet.py:23-38, dilated convs of cnn_model.py:212-240, 1x1x1 classifier of unet.UNet).
// The 3x3x3 stride-1 hot layers take the MFMA implicit-GEMM path in conv_mfma.hip instead.
//
// Roofline: these layers have arithmetic intensity of a few FLOP/byte (separable convs AI~6) and are
// HBM-bound; the design goal is one coalesced pass over x and y with weights on the scalar path.
#include "common.h"

namespace mri3d {

// ------------------------------------------------------------------ weight repack
// torch (Co, Ci, kd, kh, kw)  ->  fwd  Wf[tap][ci][coP]   (coP = Co rounded up to the cout tile)
//                             ->  dgrad Wd[tap][co][ciP]
__global__ void repack_w_fwd_kernel(const float* __restrict__ w, float* __restrict__ wp, int Co, int Ci, int taps,
                                    int CoP) {
    int total = taps * Ci * CoP;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        int co = i % CoP;
        int ci = (i / CoP) % Ci;
        int tap = i / (CoP * Ci);
        wp[i] = (co < Co) ? w[((size_t)co * Ci + ci) * taps + tap] : 0.f;
    }
}
__global__ void repack_w_dgrad_kernel(const float* __restrict__ w, float* __restrict__ wp, int Co, int Ci, int taps,
                                      int CiP) {
    int total = taps * Co * CiP;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        int ci = i % CiP;
        int co = (i / CiP) % Co;
        int tap = i / (CiP * Co);
        wp[i] = (ci < Ci) ? w[((size_t)co * Ci + ci) * taps + tap] : 0.f;
    }
}

// ------------------------------------------------------------------ forward
// One thread = one output voxel x COT output channels.  Weights are wave-uniform -> scalar loads.
template <typename T, int COT, bool VEC4>
__global__ void __launch_bounds__(256)
conv_fwd_generic_kernel(Mri3dConvGeom g, const T* __restrict__ x, const float* __restrict__ wp,
                        const float* __restrict__ bias, T* __restrict__ y, int CoP) {
    const int cot = blockIdx.y * COT;
    const int64_t nvox = (int64_t)g.n * g.dout * g.ho * g.wo;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvox; v += (int64_t)gridDim.x * blockDim.x) {
        int ow = (int)(v % g.wo);
        int64_t t = v / g.wo;
        int oh = (int)(t % g.ho);
        t /= g.ho;
        int od = (int)(t % g.dout);
        int n = (int)(t / g.dout);
        float acc[COT];
#pragma unroll
        for (int j = 0; j < COT; ++j) acc[j] = (bias != nullptr && cot + j < g.co) ? bias[cot + j] : 0.f;
        int tap = 0;
        for (int kd = 0; kd < g.kd; ++kd) {
            int id = od * g.sd - g.pd + kd * g.dd;
            for (int kh = 0; kh < g.kh; ++kh) {
                int ih = oh * g.sh - g.ph + kh * g.dh;
                for (int kw = 0; kw < g.kw; ++kw, ++tap) {
                    int iw = ow * g.sw - g.pw + kw * g.dw;
                    bool valid = (unsigned)id < (unsigned)g.di && (unsigned)ih < (unsigned)g.hi &&
                                 (unsigned)iw < (unsigned)g.wi;
                    const T* xp = x + ((((int64_t)n * g.di + (valid ? id : 0)) * g.hi + (valid ? ih : 0)) * g.wi +
                                           (valid ? iw : 0)) * g.x_ld;
                    const float* wt = wp + (size_t)tap * g.ci * CoP + cot;
                    if (VEC4) {
                        for (int ci = 0; ci < g.ci; ci += 4) {
                            float4 xv = valid ? ldf4(xp + ci) : make_float4(0, 0, 0, 0);
                            const float* w0 = wt + (size_t)ci * CoP;
#pragma unroll
                            for (int j = 0; j < COT; ++j) {
                                acc[j] = fmaf(xv.x, w0[j], acc[j]);
                                acc[j] = fmaf(xv.y, w0[CoP + j], acc[j]);
                                acc[j] = fmaf(xv.z, w0[2 * CoP + j], acc[j]);
                                acc[j] = fmaf(xv.w, w0[3 * CoP + j], acc[j]);
                            }
                        }
                    } else {
                        for (int ci = 0; ci < g.ci; ++ci) {
                            float xv = valid ? ldf(xp + ci) : 0.f;
                            const float* w0 = wt + (size_t)ci * CoP;
#pragma unroll
                            for (int j = 0; j < COT; ++j) acc[j] = fmaf(xv, w0[j], acc[j]);
                        }
                    }
                }
            }
        }
        T* yp = y + v * g.y_ld + cot;
#pragma unroll
        for (int j = 0; j < COT; ++j)
            if (cot + j < g.co) stf(yp + j, acc[j]);
    }
}

// ------------------------------------------------------------------ data gradient (also ConvTranspose3d forward)
// One thread = one input voxel x CIT input channels; gather over the output voxels that read it.
template <typename T, int CIT, bool VEC4>
__global__ void __launch_bounds__(256)
conv_dgrad_generic_kernel(Mri3dConvGeom g, const T* __restrict__ dy, const float* __restrict__ wp,
                          const float* __restrict__ bias, T* __restrict__ dx, int CiP) {
    const int cit = blockIdx.y * CIT;
    const int64_t nvox = (int64_t)g.n * g.di * g.hi * g.wi;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvox; v += (int64_t)gridDim.x * blockDim.x) {
        int iw = (int)(v % g.wi);
        int64_t t = v / g.wi;
        int ih = (int)(t % g.hi);
        t /= g.hi;
        int id = (int)(t % g.di);
        int n = (int)(t / g.di);
        float acc[CIT];
#pragma unroll
        for (int j = 0; j < CIT; ++j) acc[j] = (bias != nullptr && cit + j < g.ci) ? bias[cit + j] : 0.f;
        int tap = 0;
        for (int kd = 0; kd < g.kd; ++kd) {
            int nd = id + g.pd - kd * g.dd;
            int od = nd / g.sd;
            bool vd = nd >= 0 && (nd - od * g.sd) == 0 && od < g.dout;
            for (int kh = 0; kh < g.kh; ++kh) {
                int nh = ih + g.ph - kh * g.dh;
                int oh = nh / g.sh;
                bool vh = nh >= 0 && (nh - oh * g.sh) == 0 && oh < g.ho;
                for (int kw = 0; kw < g.kw; ++kw, ++tap) {
                    int nw = iw + g.pw - kw * g.dw;
                    int ow = nw / g.sw;
                    bool valid = vd && vh && nw >= 0 && (nw - ow * g.sw) == 0 && ow < g.wo;
                    const T* yp = dy + ((((int64_t)n * g.dout + (valid ? od : 0)) * g.ho + (valid ? oh : 0)) * g.wo +
                                            (valid ? ow : 0)) * g.y_ld;
                    const float* wt = wp + (size_t)tap * g.co * CiP + cit;
                    if (VEC4) {
                        for (int co = 0; co < g.co; co += 4) {
                            float4 gv = valid ? ldf4(yp + co) : make_float4(0, 0, 0, 0);
                            const float* w0 = wt + (size_t)co * CiP;
#pragma unroll
                            for (int j = 0; j < CIT; ++j) {
                                acc[j] = fmaf(gv.x, w0[j], acc[j]);
                                acc[j] = fmaf(gv.y, w0[CiP + j], acc[j]);
                                acc[j] = fmaf(gv.z, w0[2 * CiP + j], acc[j]);
                                acc[j] = fmaf(gv.w, w0[3 * CiP + j], acc[j]);
                            }
                        }
                    } else {
                        for (int co = 0; co < g.co; ++co) {
                            float gv = valid ? ldf(yp + co) : 0.f;
                            const float* w0 = wt + (size_t)co * CiP;
#pragma unroll
                            for (int j = 0; j < CIT; ++j) acc[j] = fmaf(gv, w0[j], acc[j]);
                        }
                    }
                }
            }
        }
        T* xp = dx + v * g.x_ld + cit;
#pragma unroll
        for (int j = 0; j < CIT; ++j)
            if (cit + j < g.ci) stf(xp + j, acc[j]);
    }
}

// ------------------------------------------------------------------ weight gradient
// grid = (voxel-chunk blocks, taps, item groups).  A block stages CH output voxels of dy and the tap-shifted x
// rows in LDS, and each thread owns a 4(ci) x 4(co) register tile of dW (or, for tiny Ci*Co, a voxel sub-group of
// it that is reduced through LDS at the end).  Per-block partials go to the workspace and are summed by
// wgrad_reduce_kernel in a fixed order (deterministic, no float atomics).
constexpr int kWgCH = 64;   // output voxels per chunk
constexpr int kWgIPT = 4;   // 4x4 items per thread (when Ci4*Co4 > 256)

template <typename T>
__global__ void __launch_bounds__(256)
conv_wgrad_generic_kernel(Mri3dConvGeom g, const T* __restrict__ x, const T* __restrict__ dy,
                          float* __restrict__ part, float* __restrict__ bias_part, int Ci4, int Co4, int nitems,
                          int vsplit) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int CiP = Ci4 * 4, CoP = Co4 * 4;
    float* dy_s = smem;                       // [CH][CoP]
    float* x_s = dy_s + kWgCH * CoP;          // [CH][CiP]
    int64_t* off_s = reinterpret_cast<int64_t*>(x_s + kWgCH * CiP);  // [CH] input voxel offset or -1
    float* red_s = reinterpret_cast<float*>(off_s + kWgCH);          // [256][16] (vsplit > 1 only)

    const int tap = blockIdx.y;
    const int kw_ = tap % g.kw, kh_ = (tap / g.kw) % g.kh, kd_ = tap / (g.kw * g.kh);
    const int tid = threadIdx.x;
    const int64_t nvox = (int64_t)g.n * g.dout * g.ho * g.wo;
    const int64_t nchunks = (nvox + kWgCH - 1) / kWgCH;

    // item ownership
    int my_item[kWgIPT];
    int vg = 0;
    if (vsplit > 1) {
        my_item[0] = tid % nitems;
        vg = tid / nitems;
        if (vg >= vsplit) my_item[0] = -1;
#pragma unroll
        for (int j = 1; j < kWgIPT; ++j) my_item[j] = -1;
    } else {
#pragma unroll
        for (int j = 0; j < kWgIPT; ++j) {
            int it = blockIdx.z * (256 * kWgIPT) + j * 256 + tid;
            my_item[j] = it < nitems ? it : -1;
        }
    }
    float acc[kWgIPT][16];
#pragma unroll
    for (int j = 0; j < kWgIPT; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    float bsum = 0.f;
    const bool do_bias = (bias_part != nullptr) && tap == 0 && blockIdx.z == 0;

    for (int64_t ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
        const int64_t v0 = ch * kWgCH;
        if (tid < kWgCH) {
            int64_t v = v0 + tid;
            int64_t off = -1;
            if (v < nvox) {
                int ow = (int)(v % g.wo);
                int64_t t = v / g.wo;
                int oh = (int)(t % g.ho);
                t /= g.ho;
                int od = (int)(t % g.dout);
                int n = (int)(t / g.dout);
                int id = od * g.sd - g.pd + kd_ * g.dd;
                int ih = oh * g.sh - g.ph + kh_ * g.dh;
                int iw = ow * g.sw - g.pw + kw_ * g.dw;
                if ((unsigned)id < (unsigned)g.di && (unsigned)ih < (unsigned)g.hi && (unsigned)iw < (unsigned)g.wi)
                    off = ((((int64_t)n * g.di + id) * g.hi + ih) * g.wi + iw) * g.x_ld;
            }
            off_s[tid] = off;
        }
        __syncthreads();
        for (int i = tid; i < kWgCH * CoP; i += 256) {
            int vv = i / CoP, c = i - vv * CoP;
            int64_t v = v0 + vv;
            dy_s[i] = (v < nvox && c < g.co) ? ldf(dy + v * g.y_ld + c) : 0.f;
        }
        for (int i = tid; i < kWgCH * CiP; i += 256) {
            int vv = i / CiP, c = i - vv * CiP;
            int64_t off = off_s[vv];
            x_s[i] = (off >= 0 && c < g.ci) ? ldf(x + off + c) : 0.f;
        }
        __syncthreads();
        if (do_bias && tid < g.co) {
            float s = 0.f;
            for (int vv = 0; vv < kWgCH; ++vv) s += dy_s[vv * CoP + tid];
            bsum += s;
        }
#pragma unroll
        for (int j = 0; j < kWgIPT; ++j) {
            if (my_item[j] < 0) continue;
            const int ci4 = my_item[j] / Co4, co4 = my_item[j] - ci4 * Co4;
            for (int vv = vg; vv < kWgCH; vv += vsplit) {
                float4 xv = *reinterpret_cast<const float4*>(x_s + vv * CiP + ci4 * 4);
                float4 gv = *reinterpret_cast<const float4*>(dy_s + vv * CoP + co4 * 4);
                acc[j][0] = fmaf(xv.x, gv.x, acc[j][0]);
                acc[j][1] = fmaf(xv.x, gv.y, acc[j][1]);
                acc[j][2] = fmaf(xv.x, gv.z, acc[j][2]);
                acc[j][3] = fmaf(xv.x, gv.w, acc[j][3]);
                acc[j][4] = fmaf(xv.y, gv.x, acc[j][4]);
                acc[j][5] = fmaf(xv.y, gv.y, acc[j][5]);
                acc[j][6] = fmaf(xv.y, gv.z, acc[j][6]);
                acc[j][7] = fmaf(xv.y, gv.w, acc[j][7]);
                acc[j][8] = fmaf(xv.z, gv.x, acc[j][8]);
                acc[j][9] = fmaf(xv.z, gv.y, acc[j][9]);
                acc[j][10] = fmaf(xv.z, gv.z, acc[j][10]);
                acc[j][11] = fmaf(xv.z, gv.w, acc[j][11]);
                acc[j][12] = fmaf(xv.w, gv.x, acc[j][12]);
                acc[j][13] = fmaf(xv.w, gv.y, acc[j][13]);
                acc[j][14] = fmaf(xv.w, gv.z, acc[j][14]);
                acc[j][15] = fmaf(xv.w, gv.w, acc[j][15]);
            }
        }
        __syncthreads();
    }

    // partial layout: part[blockIdx.x][tap][ciP][coP]
    const int taps = g.kd * g.kh * g.kw;
    float* my_part = part + ((size_t)blockIdx.x * taps + tap) * CiP * CoP;
    if (vsplit > 1) {
#pragma unroll
        for (int e = 0; e < 16; ++e) red_s[tid * 16 + e] = acc[0][e];
        __syncthreads();
        if (vg == 0 && my_item[0] >= 0) {
            const int ci4 = my_item[0] / Co4, co4 = my_item[0] - ci4 * Co4;
            float s[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) s[e] = 0.f;
            for (int q = 0; q < vsplit; ++q)
#pragma unroll
                for (int e = 0; e < 16; ++e) s[e] += red_s[(q * nitems + my_item[0]) * 16 + e];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) my_part[(ci4 * 4 + a) * CoP + co4 * 4 + b] = s[a * 4 + b];
        }
    } else {
#pragma unroll
        for (int j = 0; j < kWgIPT; ++j) {
            if (my_item[j] < 0) continue;
            const int ci4 = my_item[j] / Co4, co4 = my_item[j] - ci4 * Co4;
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) my_part[(ci4 * 4 + a) * CoP + co4 * 4 + b] = acc[j][a * 4 + b];
        }
    }
    if (do_bias && tid < g.co) bias_part[(size_t)blockIdx.x * g.co + tid] = bsum;
}

// dw[co][ci][tap] = sum_b part[b][tap][ci][co]; dbias[co] = sum_b bias_part[b][co]
// 32 lanes per output element stride over the per-block partials; lane sums are combined in a fixed order (double).
__global__ void __launch_bounds__(256)
wgrad_reduce_kernel(const float* __restrict__ part, const float* __restrict__ bias_part, float* __restrict__ dw,
                    float* __restrict__ dbias, int nb, int taps, int Ci, int Co, int CiP, int CoP) {
    __shared__ double red[256];
    const int el = threadIdx.x >> 5, ql = threadIdx.x & 31;
    const int nw = Co * Ci * taps, ntot = nw + (dbias != nullptr ? Co : 0);
    const int i = blockIdx.x * 8 + el;
    double s = 0.0;
    if (i < nw) {
        const int tap = i % taps, ci = (i / taps) % Ci, co = i / (taps * Ci);
        const float* p = part + ((size_t)tap * CiP + ci) * CoP + co;
        const size_t stride = (size_t)taps * CiP * CoP;
        for (int b = ql; b < nb; b += 32) s += (double)p[b * stride];
    } else if (i < ntot) {
        for (int b = ql; b < nb; b += 32) s += (double)bias_part[(size_t)b * Co + (i - nw)];
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if (ql == 0 && i < ntot) {
        double t = 0.0;
        for (int k = 0; k < 32; ++k) t += red[el * 32 + k];
        if (i < nw) dw[i] = (float)t;
        else dbias[i - nw] = (float)t;
    }
}

// ------------------------------------------------------------------ weight gradient, few-taps variant
// Separable (k,1,1)/(1,k,1)/(1,1,k) convs (k <= 8 taps) with narrow channels are pure HBM streams: AI ~ 6 FLOP/B.  The
// chunked kernel above spends its time on per-chunk barriers there (147 GB/s on the 6x1x1 1->8 layer).  Here a thread
// owns (voxel lane, input channel): it walks output voxels grid-stride, reads dy[v][0..8) and its x value at every tap,
// and keeps a taps x 8 block of dW in registers; lanes are combined through LDS (double) once per block, one partial
// per block in the layout wgrad_reduce_kernel expects, fixed-order final sum => deterministic.
constexpr int kSmTaps = 8, kSmCo = 8;

template <typename T>
__global__ void __launch_bounds__(256)
conv_wgrad_small_kernel(Mri3dConvGeom g, const T* __restrict__ x, const T* __restrict__ dy,
                        float* __restrict__ part, float* __restrict__ bias_part, int CiL, int CiP, int CoP) {
    __shared__ float red[256 * kSmCo];
    const int taps = g.kd * g.kh * g.kw;
    const int tid = threadIdx.x;
    const int cil = tid % CiL, vl = tid / CiL, VL = 256 / CiL;
    const int ci = blockIdx.z * CiL + cil;
    const int co0 = blockIdx.y * kSmCo;
    const bool ci_ok = ci < g.ci;
    const bool vec_dy = co0 + kSmCo <= g.co && (g.y_ld & 3) == 0 && ((uintptr_t)dy & (4 * sizeof(T) - 1)) == 0;
    float acc[kSmTaps][kSmCo], bsum[kSmCo];
#pragma unroll
    for (int t = 0; t < kSmTaps; ++t)
#pragma unroll
        for (int c = 0; c < kSmCo; ++c) acc[t][c] = 0.f;
#pragma unroll
    for (int c = 0; c < kSmCo; ++c) bsum[c] = 0.f;
    const int64_t per_n = (int64_t)g.dout * g.ho * g.wo;
    const int64_t nvox = (int64_t)g.n * per_n;
    for (int64_t v = (int64_t)blockIdx.x * VL + vl; v < nvox; v += (int64_t)gridDim.x * VL) {
        const int n = (int)(v / per_n);
        unsigned r = (unsigned)(v - (int64_t)n * per_n);
        const int ow = r % g.wo;
        r /= g.wo;
        const int oh = r % g.ho, od = r / g.ho;
        float gv[kSmCo];
        if (vec_dy) {
            const T* q = dy + v * g.y_ld + co0;
            const float4 a = ldf4(q), b = ldf4(q + 4);
            gv[0] = a.x, gv[1] = a.y, gv[2] = a.z, gv[3] = a.w, gv[4] = b.x, gv[5] = b.y, gv[6] = b.z, gv[7] = b.w;
        } else {
#pragma unroll
            for (int c = 0; c < kSmCo; ++c) gv[c] = (co0 + c < g.co) ? ldf(dy + v * g.y_ld + co0 + c) : 0.f;
        }
        if (cil == 0) {
#pragma unroll
            for (int c = 0; c < kSmCo; ++c) bsum[c] += gv[c];
        }
        const T* xn = x + (int64_t)n * g.di * g.hi * g.wi * g.x_ld + ci;
#pragma unroll
        for (int t = 0; t < kSmTaps; ++t) {
            if (t < taps) {
                const int kw = t % g.kw, kh = (t / g.kw) % g.kh, kd = t / (g.kw * g.kh);
                const int id = od * g.sd - g.pd + kd * g.dd, ih = oh * g.sh - g.ph + kh * g.dh, iw = ow * g.sw - g.pw + kw * g.dw;
                const bool ok = ci_ok && (unsigned)id < (unsigned)g.di && (unsigned)ih < (unsigned)g.hi &&
                                (unsigned)iw < (unsigned)g.wi;
                const float xv = ok ? ldf(xn + (((int64_t)id * g.hi + ih) * g.wi + iw) * g.x_ld) : 0.f;
#pragma unroll
                for (int c = 0; c < kSmCo; ++c) acc[t][c] = fmaf(xv, gv[c], acc[t][c]);
            }
        }
    }
    // combine the VL voxel lanes of every (tap, ci) through LDS, in double, in a fixed order
    for (int t = 0; t < taps; ++t) {
        __syncthreads();
#pragma unroll
        for (int c = 0; c < kSmCo; ++c) {
            float val = 0.f;
#pragma unroll
            for (int tt = 0; tt < kSmTaps; ++tt)
                if (tt == t) val = acc[tt][c];
            red[tid * kSmCo + c] = val;
        }
        __syncthreads();
        for (int o = tid; o < CiL * kSmCo; o += 256) {
            const int c = o % kSmCo, cl = o / kSmCo;
            double sdbl = 0.0;
            for (int l = 0; l < VL; ++l) sdbl += (double)red[(l * CiL + cl) * kSmCo + c];
            const int cig = blockIdx.z * CiL + cl;
            if (cig < g.ci && co0 + c < g.co)
                part[(((size_t)blockIdx.x * taps + t) * CiP + cig) * CoP + co0 + c] = (float)sdbl;
        }
    }
    if (bias_part != nullptr && blockIdx.z == 0) {
        __syncthreads();
#pragma unroll
        for (int c = 0; c < kSmCo; ++c) red[tid * kSmCo + c] = (cil == 0) ? bsum[c] : 0.f;
        __syncthreads();
        if (tid < kSmCo && co0 + tid < g.co) {
            double sdbl = 0.0;
            for (int l = 0; l < VL; ++l) sdbl += (double)red[(l * CiL) * kSmCo + tid];
            bias_part[(size_t)blockIdx.x * g.co + co0 + tid] = (float)sdbl;
        }
    }
}

static bool wgrad_small_ok(const Mri3dConvGeom& g) {
    const int taps = g.kd * g.kh * g.kw;
    return taps <= kSmTaps && g.ci <= 64 && g.co <= 64 && (int64_t)g.dout * g.ho * g.wo < 0x7fffffffLL;
}

struct WgradSmallPlan { int CiL, gz, gy, gx, CiP, CoP; size_t part_floats, bias_floats; };
static WgradSmallPlan wgrad_small_plan(const Mri3dConvGeom& g) {
    WgradSmallPlan p;
    int cil = 1;
    while (cil < g.ci && cil < 16) cil <<= 1;
    p.CiL = cil;
    p.gz = cdiv(g.ci, cil);
    p.gy = cdiv(g.co, kSmCo);
    p.CiP = cdiv(g.ci, 4) * 4;
    p.CoP = cdiv(g.co, 4) * 4;
    const int64_t nvox = (int64_t)g.n * g.dout * g.ho * g.wo;
    int64_t want = cdiv64(nvox, (int64_t)(256 / cil) * 8);
    int cap = 2048 / (p.gy * p.gz);
    if (cap < 8) cap = 8;
    p.gx = (int)std::max<int64_t>(1, std::min<int64_t>(want, cap));
    p.part_floats = (size_t)p.gx * g.kd * g.kh * g.kw * p.CiP * p.CoP;
    p.bias_floats = (size_t)p.gx * g.co;
    return p;
}

// ------------------------------------------------------------------ host side
static int pick_tile(int c) { return c >= 16 ? 16 : (c > 4 ? 8 : (c > 2 ? 4 : 2)); }

struct WgradPlan {
    int Ci4, Co4, nitems, vsplit, gx, gz, taps;
    size_t part_floats, bias_floats, smem;
};
static WgradPlan wgrad_plan(const Mri3dConvGeom& g) {
    WgradPlan p;
    p.Ci4 = cdiv(g.ci, 4);
    p.Co4 = cdiv(g.co, 4);
    p.nitems = p.Ci4 * p.Co4;
    p.taps = g.kd * g.kh * g.kw;
    p.vsplit = p.nitems >= 256 ? 1 : (256 / p.nitems);
    if (p.vsplit > kWgCH) p.vsplit = kWgCH;
    p.gz = p.vsplit > 1 ? 1 : cdiv(p.nitems, 256 * kWgIPT);
    int64_t nvox = (int64_t)g.n * g.dout * g.ho * g.wo;
    int64_t nchunks = cdiv64(nvox, kWgCH);
    int gx = 2048 / (p.taps * p.gz);
    if (gx < 8) gx = 8;
    if (gx > 256) gx = 256;
    if (gx > nchunks) gx = (int)nchunks;
    if (gx < 1) gx = 1;
    p.gx = gx;
    p.part_floats = (size_t)gx * p.taps * p.Ci4 * 4 * p.Co4 * 4;
    p.bias_floats = (size_t)gx * g.co;
    p.smem = (size_t)kWgCH * (p.Ci4 + p.Co4) * 4 * sizeof(float) + kWgCH * sizeof(int64_t) +
             (p.vsplit > 1 ? 256 * 16 * sizeof(float) : 0);
    return p;
}

size_t conv_generic_workspace_bytes(const Mri3dConvGeom& g, int pass) {
    const int taps = g.kd * g.kh * g.kw;
    if (pass == MRI3D_PASS_FWD) {
        int t = pick_tile(g.co);
        return (size_t)taps * g.ci * cdiv(g.co, t) * t * sizeof(float);
    }
    if (pass == MRI3D_PASS_DGRAD) {
        int t = pick_tile(g.ci);
        return (size_t)taps * g.co * cdiv(g.ci, t) * t * sizeof(float);
    }
    WgradPlan p = wgrad_plan(g);
    size_t a = (p.part_floats + p.bias_floats) * sizeof(float);
    if (wgrad_small_ok(g)) {
        WgradSmallPlan q = wgrad_small_plan(g);
        a = std::max(a, (q.part_floats + q.bias_floats) * sizeof(float));
    }
    return a;
}

template <int TL>
static void launch_fwd(const Mri3dConvGeom& g, const void* x, const float* wp, const float* bias, void* y, int CoP,
                       hipStream_t s) {
    int64_t nvox = (int64_t)g.n * g.dout * g.ho * g.wo;
    dim3 grid((unsigned)std::min<int64_t>(cdiv64(nvox, 256), 8192), CoP / TL);
    bool vec = (g.ci % 4 == 0) && (g.x_ld % 4 == 0) && aligned_vec4(g.dtype, x);
    MRI3D_DISPATCH_DTYPE(g.dtype, T, {
        if (vec)
            hipLaunchKernelGGL((conv_fwd_generic_kernel<T, TL, true>), grid, dim3(256), 0, s, g, (const T*)x, wp, bias, (T*)y, CoP);
        else
            hipLaunchKernelGGL((conv_fwd_generic_kernel<T, TL, false>), grid, dim3(256), 0, s, g, (const T*)x, wp, bias, (T*)y, CoP);
    });
}

int conv_generic_fwd(const Mri3dConvGeom& g, const void* x, const float* w, const float* bias, void* y, void* ws,
                     size_t ws_bytes, hipStream_t s) {
    const int taps = g.kd * g.kh * g.kw;
    const int TL = pick_tile(g.co);
    const int CoP = cdiv(g.co, TL) * TL;
    size_t need = (size_t)taps * g.ci * CoP * sizeof(float);
    MRI3D_REQUIRE(ws != nullptr && ws_bytes >= need, MRI3D_EWORKSPACE, "conv3d_fwd: workspace %zu < %zu", ws_bytes, need);
    float* wp = static_cast<float*>(ws);
    int total = taps * g.ci * CoP;
    hipLaunchKernelGGL(repack_w_fwd_kernel, dim3(std::min(cdiv(total, 256), 1024)), dim3(256), 0, s, w, wp, g.co, g.ci,
                       taps, CoP);
    switch (TL) {
        case 16: launch_fwd<16>(g, x, wp, bias, y, CoP, s); break;
        case 8: launch_fwd<8>(g, x, wp, bias, y, CoP, s); break;
        case 4: launch_fwd<4>(g, x, wp, bias, y, CoP, s); break;
        default: launch_fwd<2>(g, x, wp, bias, y, CoP, s); break;
    }
    return check_launch("conv3d_fwd(generic)");
}

template <int TL>
static void launch_dgrad(const Mri3dConvGeom& g, const void* dy, const float* wp, const float* bias, void* dx,
                         int CiP, hipStream_t s) {
    int64_t nvox = (int64_t)g.n * g.di * g.hi * g.wi;
    dim3 grid((unsigned)std::min<int64_t>(cdiv64(nvox, 256), 8192), CiP / TL);
    bool vec = (g.co % 4 == 0) && (g.y_ld % 4 == 0) && aligned_vec4(g.dtype, dy);
    MRI3D_DISPATCH_DTYPE(g.dtype, T, {
        if (vec)
            hipLaunchKernelGGL((conv_dgrad_generic_kernel<T, TL, true>), grid, dim3(256), 0, s, g, (const T*)dy, wp, bias, (T*)dx, CiP);
        else
            hipLaunchKernelGGL((conv_dgrad_generic_kernel<T, TL, false>), grid, dim3(256), 0, s, g, (const T*)dy, wp, bias, (T*)dx, CiP);
    });
}

int conv_generic_dgrad(const Mri3dConvGeom& g, const void* dy, const float* w, const float* bias, void* dx, void* ws,
                       size_t ws_bytes, hipStream_t s) {
    const int taps = g.kd * g.kh * g.kw;
    const int TL = pick_tile(g.ci);
    const int CiP = cdiv(g.ci, TL) * TL;
    size_t need = (size_t)taps * g.co * CiP * sizeof(float);
    MRI3D_REQUIRE(ws != nullptr && ws_bytes >= need, MRI3D_EWORKSPACE, "conv3d_dgrad: workspace %zu < %zu", ws_bytes,
                  need);
    float* wp = static_cast<float*>(ws);
    int total = taps * g.co * CiP;
    hipLaunchKernelGGL(repack_w_dgrad_kernel, dim3(std::min(cdiv(total, 256), 1024)), dim3(256), 0, s, w, wp, g.co,
                       g.ci, taps, CiP);
    switch (TL) {
        case 16: launch_dgrad<16>(g, dy, wp, bias, dx, CiP, s); break;
        case 8: launch_dgrad<8>(g, dy, wp, bias, dx, CiP, s); break;
        case 4: launch_dgrad<4>(g, dy, wp, bias, dx, CiP, s); break;
        default: launch_dgrad<2>(g, dy, wp, bias, dx, CiP, s); break;
    }
    return check_launch("conv3d_dgrad(generic)");
}

int conv_generic_wgrad(const Mri3dConvGeom& g, const void* x, const void* dy, float* dw, float* dbias, void* ws,
                       size_t ws_bytes, hipStream_t s) {
    if (wgrad_small_ok(g)) {
        WgradSmallPlan q = wgrad_small_plan(g);
        const size_t need = (q.part_floats + q.bias_floats) * sizeof(float);
        MRI3D_REQUIRE(ws != nullptr && ws_bytes >= need, MRI3D_EWORKSPACE, "conv3d_wgrad: workspace %zu < %zu", ws_bytes, need);
        float* part = static_cast<float*>(ws);
        float* bias_part = dbias ? part + q.part_floats : nullptr;
        const int taps = g.kd * g.kh * g.kw;
        // partial slots of padded channels are never written by the kernel and never read by the reduce
        MRI3D_DISPATCH_DTYPE(g.dtype, T, {
            hipLaunchKernelGGL(conv_wgrad_small_kernel<T>, dim3(q.gx, q.gy, q.gz), dim3(256), 0, s, g, (const T*)x,
                               (const T*)dy, part, bias_part, q.CiL, q.CiP, q.CoP);
        });
        const int total = g.co * g.ci * taps;
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(cdiv(total + g.co, 8)), dim3(256), 0, s, part, bias_part, dw,
                           dbias, q.gx, taps, g.ci, g.co, q.CiP, q.CoP);
        return check_launch("conv3d_wgrad(small)");
    }
    WgradPlan p = wgrad_plan(g);
    size_t need = (p.part_floats + p.bias_floats) * sizeof(float);
    MRI3D_REQUIRE(ws != nullptr && ws_bytes >= need, MRI3D_EWORKSPACE, "conv3d_wgrad: workspace %zu < %zu", ws_bytes,
                  need);
    MRI3D_REQUIRE(p.smem <= 160 * 1024, MRI3D_ENOTSUP, "conv3d_wgrad: Ci=%d Co=%d needs %zu B of LDS", g.ci, g.co,
                  p.smem);
    float* part = static_cast<float*>(ws);
    float* bias_part = dbias ? part + p.part_floats : nullptr;
    MRI3D_DISPATCH_DTYPE(g.dtype, T, {
        if (p.smem > 64 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_generic_kernel<T>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.smem);
        hipLaunchKernelGGL(conv_wgrad_generic_kernel<T>, dim3(p.gx, p.taps, p.gz), dim3(256), p.smem, s, g, (const T*)x,
                           (const T*)dy, part, bias_part, p.Ci4, p.Co4, p.nitems, p.vsplit);
    });
    int total = g.co * g.ci * p.taps;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(cdiv(total + g.co, 8)), dim3(256), 0, s, part, bias_part, dw,
                       dbias, p.gx, p.taps, g.ci, g.co, p.Ci4 * 4, p.Co4 * 4);
    return check_launch("conv3d_wgrad(generic)");
}

}  // namespace mri3d
