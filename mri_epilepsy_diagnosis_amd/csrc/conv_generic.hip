// conv_generic.hip — direct NDHWC Conv3d forward / data-gradient / weight-gradient for ANY kernel size,
// stride, padding and dilation (separable (k,1,1) convs of AE_model.py:9-26, strided convs of
// modified_3dunet.py:23-38, dilated convs of cnn_model.py:212-240, 1x1x1 classifier of unet.UNet).
// The 3x3x3 stride-1 hot layers take the MFMA implicit-GEMM path in conv_mfma.hip instead.
//
// Roofline: these layers have arithmetic intensity of a few FLOP/byte (separable convs AI~6) and are
// HBM-bound; the design goal is one coalesced pass over x and y with weights on the scalar path.
#include "common.h"

namespace mri3d {

// sum over the 64 lanes of a wave, the same value in every lane: xor-1 / xor-2 inside quads, half-mirror and mirror inside
// the 16-lane DPP rows, then the four row sums through readlane
__device__ __forceinline__ float wave_sum_dpp(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));   // row_mirror
    const int vi = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 0));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 32));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 48));
    return (r0 + r1) + (r2 + r3);
}

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

// ------------------------------------------------------------------ forward, few taps (separable convs of the autoencoder)
// Same arithmetic and thread = (output voxel, COT output channels) as the kernel above, for filters of <= 8 taps on input channels in
// multiples of 4 (AE_model.py:9-26, 74-91: (6,1,1)/(1,6,1)/(1,1,6) stride 2 and (3,1,1)/(1,3,1)/(1,1,3)).  There every tap's load
// sits inside nested run-time loops behind `valid ? load : 0`, so it gets its own wait (12 serial round trips per voxel of the 8 -> 8
// layer) and each voxel pays three 64-bit divisions.  Here voxels are walked slab-wise with 32-bit arithmetic and, per input-channel
// quad, the loads of ALL taps are issued together (clamped address, masked afterwards) before their FMAs.
// CV = input channels per load: 4 (Ci % 4 == 0, 16-byte loads) or 1 (any Ci; the autoencoder's first conv has Ci = 1)
template <typename T, int COT, int NTAPS, int CV>
__global__ void __launch_bounds__(256)
conv_fwd_taps_kernel(Mri3dConvGeom g, const T* __restrict__ x, const float* __restrict__ wp, const float* __restrict__ bias,
                     T* __restrict__ y, int CoP, int hch) {
    const int taps = g.kd * g.kh * g.kw;
    const int cot = blockIdx.y * COT;
    int tdd[NTAPS], tdh[NTAPS], tdw[NTAPS];   // tap offsets in voxels per axis, and as one element offset (all wave-uniform)
    int64_t tdelta[NTAPS];
#pragma unroll
    for (int t = 0; t < NTAPS; ++t) {
        const int tt = t < taps ? t : 0;
        tdw[t] = (tt % g.kw) * g.dw, tdh[t] = ((tt / g.kw) % g.kh) * g.dh, tdd[t] = (tt / (g.kw * g.kh)) * g.dd;
        tdelta[t] = (((int64_t)tdd[t] * g.hi + tdh[t]) * g.wi + tdw[t]) * g.x_ld;
    }
    const int hchunks = (g.ho + hch - 1) / hch;
    const int slabs = g.n * g.dout * hchunks;
    for (int slab = blockIdx.x; slab < slabs; slab += gridDim.x) {
        const int hc = slab % hchunks, nd = slab / hchunks;
        const int n = nd / g.dout, od = nd - n * g.dout;
        const int h0 = hc * hch, hn = min(hch, g.ho - h0);
        const unsigned inner = (unsigned)hn * g.wo;
        const T* xn = x + (int64_t)n * g.di * g.hi * g.wi * g.x_ld;
        T* yn = y + (((int64_t)nd * g.ho + h0) * g.wo) * g.y_ld + cot;
        for (unsigned e = threadIdx.x; e < inner; e += blockDim.x) {
            const int ow = e % g.wo, oh = h0 + e / g.wo;
            // a tap's address = the voxel's base offset + a wave-uniform delta (one add); a tap outside the volume re-reads a valid
            // element (`safe`: the sample's first voxel) and is masked
            const int id0 = od * g.sd - g.pd, ih0 = oh * g.sh - g.ph, iw0 = ow * g.sw - g.pw;
            const int64_t base = (((int64_t)id0 * g.hi + ih0) * g.wi + iw0) * g.x_ld;
            int64_t off[NTAPS];
            unsigned okm = 0;
#pragma unroll
            for (int t = 0; t < NTAPS; ++t) {
                const bool ok = t < taps && (unsigned)(id0 + tdd[t]) < (unsigned)g.di && (unsigned)(ih0 + tdh[t]) < (unsigned)g.hi &&
                                (unsigned)(iw0 + tdw[t]) < (unsigned)g.wi;
                okm |= ok ? (1u << t) : 0u;
                off[t] = ok ? base + tdelta[t] : 0;
            }
            float acc[COT];
#pragma unroll
            for (int j = 0; j < COT; ++j) acc[j] = (bias != nullptr && cot + j < g.co) ? bias[cot + j] : 0.f;
            for (int ci = 0; ci < g.ci; ci += CV) {
                float xv[NTAPS][CV];
#pragma unroll
                for (int t = 0; t < NTAPS; ++t) {
                    if constexpr (CV == 4) {
                        const float4 q = ldf4(xn + off[t] + ci);
                        xv[t][0] = q.x, xv[t][1] = q.y, xv[t][2] = q.z, xv[t][3] = q.w;
                    } else {
                        xv[t][0] = ldf(xn + off[t] + ci);
                    }
                }
#pragma unroll
                for (int t = 0; t < NTAPS; ++t) {
                    if (t < taps) {   // uniform
                        const bool ok = (okm >> t) & 1u;
                        const float* w0 = wp + ((size_t)t * g.ci + ci) * CoP + cot;
#pragma unroll
                        for (int c = 0; c < CV; ++c) {
                            const float xc = ok ? xv[t][c] : 0.f;
#pragma unroll
                            for (int j = 0; j < COT; ++j) acc[j] = fmaf(xc, w0[c * CoP + j], acc[j]);
                        }
                    }
                }
            }
            T* yp = yn + (int64_t)e * g.y_ld;
            if (COT % 4 == 0 && cot + COT <= g.co && (g.y_ld & 3) == 0) {
#pragma unroll
                for (int j = 0; j < COT; j += 4) stf4(yp + j, make_float4(acc[j], acc[j + 1], acc[j + 2], acc[j + 3]));
            } else {
#pragma unroll
                for (int j = 0; j < COT; ++j)
                    if (cot + j < g.co) stf(yp + j, acc[j]);
            }
        }
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

// ------------------------------------------------------------------ data gradient, few taps (separable convs of the autoencoder)
// The gather above with slab-wise 32-bit voxel arithmetic and, per output-channel quad, the dy loads of ALL taps issued together
// (clamped address, masked afterwards: taps that do not reach this input voxel — wrong stride residue, outside the volume — cost a
// cached load instead of a branch with its own wait).  <= 8 taps, Co in multiples of 4.
// CV = output channels per dy load: 4 (Co % 4 == 0) or 1 (Co = 1: the 8 -> 1 convs in front of the autoencoder's last block)
template <typename T, int CIT, int NTAPS, int CV>
__global__ void __launch_bounds__(256)
conv_dgrad_taps_kernel(Mri3dConvGeom g, const T* __restrict__ dy, const float* __restrict__ wp, const float* __restrict__ bias,
                       T* __restrict__ dx, int CiP, int hch) {
    const int taps = g.kd * g.kh * g.kw;
    const int cit = blockIdx.y * CIT;
    int tkd[NTAPS], tkh[NTAPS], tkw[NTAPS];
#pragma unroll
    for (int t = 0; t < NTAPS; ++t) {
        const int tt = t < taps ? t : 0;
        tkw[t] = tt % g.kw, tkh[t] = (tt / g.kw) % g.kh, tkd[t] = tt / (g.kw * g.kh);
    }
    const int hchunks = (g.hi + hch - 1) / hch;
    const int slabs = g.n * g.di * hchunks;
    for (int slab = blockIdx.x; slab < slabs; slab += gridDim.x) {
        const int hc = slab % hchunks, nd_ = slab / hchunks;
        const int n = nd_ / g.di, id = nd_ - n * g.di;
        const int h0 = hc * hch, hn = min(hch, g.hi - h0);
        const unsigned inner = (unsigned)hn * g.wi;
        const T* yn = dy + (int64_t)n * g.dout * g.ho * g.wo * g.y_ld;
        T* xn = dx + (((int64_t)nd_ * g.hi + h0) * g.wi) * g.x_ld + cit;
        for (unsigned e = threadIdx.x; e < inner; e += blockDim.x) {
            const int iw = e % g.wi, ih = h0 + e / g.wi;   // (incremental counters instead of this division pair: measured, no gain)
            int64_t off[NTAPS];
            unsigned okm = 0;
#pragma unroll
            for (int t = 0; t < NTAPS; ++t) {
                // (launched for unit strides only — launch_dgrad — so the output voxel of a tap is the shifted input voxel: no
                // run-time integer divisions, which were most of this kernel's instructions)
                const int od = id + g.pd - tkd[t] * g.dd, oh = ih + g.ph - tkh[t] * g.dh, ow = iw + g.pw - tkw[t] * g.dw;
                const bool ok = t < taps && (unsigned)od < (unsigned)g.dout && (unsigned)oh < (unsigned)g.ho && (unsigned)ow < (unsigned)g.wo;
                okm |= ok ? (1u << t) : 0u;
                const int cd = min(max(od, 0), g.dout - 1), chh = min(max(oh, 0), g.ho - 1), cw = min(max(ow, 0), g.wo - 1);
                off[t] = (((int64_t)cd * g.ho + chh) * g.wo + cw) * g.y_ld;
            }
            float acc[CIT];
#pragma unroll
            for (int j = 0; j < CIT; ++j) acc[j] = (bias != nullptr && cit + j < g.ci) ? bias[cit + j] : 0.f;
            for (int co = 0; co < g.co; co += CV) {
                float gv[NTAPS][CV];
#pragma unroll
                for (int t = 0; t < NTAPS; ++t) {
                    if constexpr (CV == 4) {
                        const float4 q = ldf4(yn + off[t] + co);
                        gv[t][0] = q.x, gv[t][1] = q.y, gv[t][2] = q.z, gv[t][3] = q.w;
                    } else {
                        gv[t][0] = ldf(yn + off[t] + co);
                    }
                }
#pragma unroll
                for (int t = 0; t < NTAPS; ++t) {
                    if (t < taps) {   // uniform
                        const bool ok = (okm >> t) & 1u;
                        const float* w0 = wp + ((size_t)t * g.co + co) * CiP + cit;
#pragma unroll
                        for (int c = 0; c < CV; ++c) {
                            const float gc = ok ? gv[t][c] : 0.f;
#pragma unroll
                            for (int j = 0; j < CIT; ++j) acc[j] = fmaf(gc, w0[c * CiP + j], acc[j]);
                        }
                    }
                }
            }
            T* xp = xn + (int64_t)e * g.x_ld;
            if (CIT % 4 == 0 && cit + CIT <= g.ci && (g.x_ld & 3) == 0) {
#pragma unroll
                for (int j = 0; j < CIT; j += 4) stf4(xp + j, make_float4(acc[j], acc[j + 1], acc[j + 2], acc[j + 3]));
            } else {
#pragma unroll
                for (int j = 0; j < CIT; ++j)
                    if (cit + j < g.ci) stf(xp + j, acc[j]);
            }
        }
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

// ------------------------------------------------------------------ weight gradient, few taps, channel QUADS
// The encoder's separable convs (AE_model.py:9-26: (6,1,1)/(1,6,1)/(1,1,6) stride 2, 8 -> 8 ... 32 -> 32) have Ci, Co in multiples
// of 4.  In the per-channel kernel above the 8 lanes of a voxel each fetch ONE 4-byte x value per tap and all re-read the same 32
// bytes of dy: 8 load instructions per lane for 56 bytes (0.84 TB/s on 8 -> 8 (1,6,1) at 80x96x160).  Here a lane owns (voxel,
// input-channel quad, output-channel quad): one 16-byte x load per tap, one 16-byte dy load, a taps x 4 x 4 register block.
// Voxels are walked slab-wise — (n, od, h-chunk) per workgroup iteration, 32-bit arithmetic per element — instead of decoding a
// 64-bit flat index per voxel.  Lanes are combined through LDS (double) once per workgroup: same partial layout and final
// fixed-order sum as above (deterministic).
// NTAPS = 4, 6 or 8 accumulator slots (the smallest that holds the filter: fewer registers, more resident waves to cover the latency)
// CIV = input channels per lane: 4 (Ci % 4 == 0) or 1 (Ci = 1: the autoencoder's first conv, (6,1,1) 1 -> 8)
template <typename T, int NTAPS, int CIV>
__global__ void __launch_bounds__(256)
conv_wgrad_quads_kernel(Mri3dConvGeom g, const T* __restrict__ x, const T* __restrict__ dy, float* __restrict__ part,
                        float* __restrict__ bias_part, int QI, int QO, int hch, int CiP, int CoP) {
    __shared__ float red[256 * 16];
    const int taps = g.kd * g.kh * g.kw;
    const int tid = threadIdx.x;
    const int PV = QI * QO, VL = 256 / PV;           // lanes per voxel, voxel lanes per workgroup (PV divides 256: host)
    const int pq = tid % PV, vl = tid / PV;
    const int iq = pq % QI, oq = pq / QI;
    float acc[NTAPS][CIV][4], bsum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < NTAPS; ++t)
#pragma unroll
        for (int a = 0; a < CIV; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[t][a][b] = 0.f;
    // tap offsets in voxels per axis and as ONE element offset (all wave-uniform): a tap's address is the voxel's base offset plus
    // a scalar — no clamps, no multiplications per tap (they were 230 of the 330 instructions per voxel of the 8 -> 8 layer)
    int tdd[NTAPS], tdh[NTAPS], tdw[NTAPS];
    int64_t tdelta[NTAPS];
#pragma unroll
    for (int t = 0; t < NTAPS; ++t) {
        const int tt = t < taps ? t : 0;
        tdw[t] = (tt % g.kw) * g.dw, tdh[t] = ((tt / g.kw) % g.kh) * g.dh, tdd[t] = (tt / (g.kw * g.kh)) * g.dd;
        tdelta[t] = (((int64_t)tdd[t] * g.hi + tdh[t]) * g.wi + tdw[t]) * g.x_ld;
    }
    const int hchunks = (g.ho + hch - 1) / hch;
    const int slabs = g.n * g.dout * hchunks;
    for (int slab = blockIdx.x; slab < slabs; slab += gridDim.x) {
        const int hc = slab % hchunks, nd = slab / hchunks;
        const int n = nd / g.dout, od = nd - n * g.dout;
        const int h0 = hc * hch, hn = min(hch, g.ho - h0);
        const unsigned inner = (unsigned)hn * g.wo;
        const T* xn = x + (int64_t)n * g.di * g.hi * g.wi * g.x_ld + CIV * iq;
        const T* yn = dy + (((int64_t)nd * g.ho + h0) * g.wo) * g.y_ld + 4 * oq;
        // ALL tap loads of a voxel first, unconditional (clamped address, masked afterwards; taps past the filter re-read tap 0):
        // a load inside `if (ok)` / `if (t < taps)` is followed by its own s_waitcnt and the taps become serial round trips to
        // L2 (7 per voxel: 0.51 ms on the (1,6,1) 8 -> 8 layer at 80x96x160, waves waiting 73 % of their cycles; batched 0.31 ms).
        auto fetch = [&](unsigned e, float4& gv, float (&xv)[NTAPS][CIV], unsigned& okm) {
            const bool live = e < inner;
            const unsigned ec = live ? e : inner - 1;
            const int ow = ec % g.wo, oh = h0 + ec / g.wo;
            gv = ldf4(yn + (int64_t)ec * g.y_ld);
            okm = 0;
            const int id0 = od * g.sd - g.pd, ih0 = oh * g.sh - g.ph, iw0 = ow * g.sw - g.pw;
            const int64_t base = (((int64_t)id0 * g.hi + ih0) * g.wi + iw0) * g.x_ld;
#pragma unroll
            for (int t = 0; t < NTAPS; ++t) {
                const bool ok = live && t < taps && (unsigned)(id0 + tdd[t]) < (unsigned)g.di && (unsigned)(ih0 + tdh[t]) < (unsigned)g.hi &&
                                (unsigned)(iw0 + tdw[t]) < (unsigned)g.wi;
                okm |= ok ? (1u << t) : 0u;
                const T* xp = xn + (ok ? base + tdelta[t] : 0);   // a tap outside the volume re-reads the sample's first voxel and is masked
                if constexpr (CIV == 4) {
                    const float4 q4 = ldf4(xp);
                    xv[t][0] = q4.x, xv[t][1] = q4.y, xv[t][2] = q4.z, xv[t][3] = q4.w;
                } else {
                    xv[t][0] = ldf(xp);
                }
            }
            if (!live) gv = make_float4(0.f, 0.f, 0.f, 0.f);
        };
        auto consume = [&](const float4& gv, const float (&xv)[NTAPS][CIV], unsigned okm) {
            const float gq[4] = {gv.x, gv.y, gv.z, gv.w};
            if (iq == 0) {
#pragma unroll
                for (int b = 0; b < 4; ++b) bsum[b] += gq[b];
            }
#pragma unroll
            for (int t = 0; t < NTAPS; ++t) {
                const bool ok = (okm >> t) & 1u;
#pragma unroll
                for (int a = 0; a < CIV; ++a) {
                    const float xa = ok ? xv[t][a] : 0.f;
#pragma unroll
                    for (int b = 0; b < 4; ++b) acc[t][a][b] = fmaf(xa, gq[b], acc[t][a][b]);
                }
            }
        };
        // (a second register set for the next voxel's loads was measured: with 128 accumulators it pushes the kernel to 256 VGPRs
        //  and 80 spilled SGPRs and is slower, 0.50 vs 0.31 ms; fewer accumulator slots and more waves hide the latency instead)
        for (unsigned e = vl; e < inner; e += VL) {
            float4 gv;
            float xv[NTAPS][CIV];
            unsigned okm;
            fetch(e, gv, xv, okm);
            consume(gv, xv, okm);
        }
    }
    // combine the VL voxel lanes of every (tap, ci, co) through LDS, in double, in a fixed order
    for (int t = 0; t < taps; ++t) {
        __syncthreads();
#pragma unroll
        for (int a = 0; a < CIV; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                float val = 0.f;
#pragma unroll
                for (int tt = 0; tt < NTAPS; ++tt)
                    if (tt == t) val = acc[tt][a][b];
                red[tid * 16 + a * 4 + b] = val;
            }
        __syncthreads();
        for (int o = tid; o < PV * (CIV * 4); o += 256) {
            const int ab = o % (CIV * 4), q = o / (CIV * 4);
            double sdbl = 0.0;
            for (int l = 0; l < VL; ++l) sdbl += (double)red[(l * PV + q) * 16 + ab];
            const int ci = (q % QI) * CIV + ab / 4, co = (q / QI) * 4 + ab % 4;
            if (ci < g.ci && co < g.co) part[(((size_t)blockIdx.x * taps + t) * CiP + ci) * CoP + co] = (float)sdbl;
        }
    }
    if (bias_part != nullptr) {
        __syncthreads();
#pragma unroll
        for (int b = 0; b < 4; ++b) red[tid * 4 + b] = (iq == 0) ? bsum[b] : 0.f;
        __syncthreads();
        if (tid < g.co) {
            double sdbl = 0.0;
            for (int l = 0; l < VL; ++l) sdbl += (double)red[((l * PV) + (tid / 4) * QI) * 4 + (tid % 4)];
            bias_part[(size_t)blockIdx.x * g.co + tid] = (float)sdbl;
        }
    }
}

// ------------------------------------------------------------------ weight gradient, few taps, ONE output channel
// The decoder's last separable convs end in a single channel (8 -> 1 and 1 -> 1 with (3,1,1)/(1,3,1)/(1,1,3) kernels at full
// resolution, AE_model.py:110-160).  In the kernel above a lane owns one input channel and reads it 4 bytes at a time while 7 of
// its 8 output-channel slots stay empty (1.32 ms for 8 -> 1 on 4 x 160x192x160, 536 GB/s).  Here a lane owns a VOXEL: one dy
// value, CI input channels per tap as 16-byte loads, taps x CI accumulators; lanes are combined with DPP wave sums.
template <typename T, int CI>
__global__ void __launch_bounds__(256)
conv_wgrad_co1_kernel(Mri3dConvGeom g, const T* __restrict__ x, const T* __restrict__ dy, float* __restrict__ part,
                      float* __restrict__ bias_part) {
    __shared__ float red[4][kSmTaps * CI + 1];
    const int taps = g.kd * g.kh * g.kw;
    const int tid = threadIdx.x;
    float acc[kSmTaps][CI], bsum = 0.f;
#pragma unroll
    for (int t = 0; t < kSmTaps; ++t)
#pragma unroll
        for (int c = 0; c < CI; ++c) acc[t][c] = 0.f;
    const int64_t per_n = (int64_t)g.dout * g.ho * g.wo;
    const int64_t nvox = (int64_t)g.n * per_n;
    for (int64_t v = (int64_t)blockIdx.x * 256 + tid; v < nvox; v += (int64_t)gridDim.x * 256) {
        const int n = (int)(v / per_n);
        unsigned r = (unsigned)(v - (int64_t)n * per_n);
        const int ow = r % g.wo;
        r /= g.wo;
        const int oh = r % g.ho, od = r / g.ho;
        const float gv = ldf(dy + v * g.y_ld);
        bsum += gv;
        const T* xn = x + (int64_t)n * g.di * g.hi * g.wi * g.x_ld;
#pragma unroll
        for (int t = 0; t < kSmTaps; ++t) {
            if (t < taps) {
                const int kw = t % g.kw, kh = (t / g.kw) % g.kh, kd = t / (g.kw * g.kh);
                const int id = od * g.sd - g.pd + kd * g.dd, ih = oh * g.sh - g.ph + kh * g.dh, iw = ow * g.sw - g.pw + kw * g.dw;
                const bool ok = (unsigned)id < (unsigned)g.di && (unsigned)ih < (unsigned)g.hi && (unsigned)iw < (unsigned)g.wi;
                const T* xp = xn + (((int64_t)min(max(id, 0), g.di - 1) * g.hi + min(max(ih, 0), g.hi - 1)) * g.wi +
                                    min(max(iw, 0), g.wi - 1)) * g.x_ld;
                const float gm = ok ? gv : 0.f;
                if constexpr (CI == 1) {
                    acc[t][0] = fmaf(ldf(xp), gm, acc[t][0]);
                } else {
#pragma unroll
                    for (int c = 0; c < CI; c += 4) {
                        const float4 q = ldf4(xp + c);
                        acc[t][c] = fmaf(q.x, gm, acc[t][c]);
                        acc[t][c + 1] = fmaf(q.y, gm, acc[t][c + 1]);
                        acc[t][c + 2] = fmaf(q.z, gm, acc[t][c + 2]);
                        acc[t][c + 3] = fmaf(q.w, gm, acc[t][c + 3]);
                    }
                }
            }
        }
    }
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int t = 0; t < kSmTaps; ++t)
#pragma unroll
        for (int c = 0; c < CI; ++c) {
            const float vsum = wave_sum_dpp(acc[t][c]);
            if (lane == 0) red[wave][t * CI + c] = vsum;
        }
    {
        const float vsum = wave_sum_dpp(bsum);
        if (lane == 0) red[wave][kSmTaps * CI] = vsum;
    }
    __syncthreads();
    if (tid < taps * CI)   // part[blk][tap][ci][co = 0]
        part[(size_t)blockIdx.x * taps * CI + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
    else if (tid == kSmTaps * CI && bias_part)
        bias_part[blockIdx.x] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
}

constexpr int kCo1Blocks = 1024;

static bool wgrad_co1_ok(const Mri3dConvGeom& g) {
    static const int off = tuning_knob("MRI3D_CO1_OFF", 0);   // tuning aid (A/B)
    return !off && g.co == 1 && g.kd * g.kh * g.kw <= kSmTaps && (g.ci == 1 || ((g.ci == 4 || g.ci == 8 || g.ci == 16) && g.x_ld % 4 == 0)) &&
           (int64_t)g.dout * g.ho * g.wo < 0x7fffffffLL;
}

// ------------------------------------------------------------------ first layer: 3x3x3, stride 1, pad 1, ONE input channel
// unet.UNet / Modified3DUNet / CNN all start with Conv3d(1, 8|16, 3, padding=1) on the whole volume.  With Cin = 1 there is
// no GEMM to speak of (K = 27): forward reads 4 B and writes 4*Co B per voxel, the weight gradient reads both, and 27*Co
// FMAs per voxel keep the VALUs about as busy as HBM (~0.05 ms each per 2 x 160x192x160 volumes).  The generic forward
// kernel needs 0.39 ms there and the CK = 1 MFMA weight gradient 0.47 ms; these two take 0.18 and 0.25 ms:
//   forward: a 4 x 8 x (8*VPT) voxel tile of x (+halo) sits in LDS, a lane owns VPT consecutive voxels along W and all Co
//            outputs; per (kd, kh) it reads its VPT+2 inputs once and reuses them for the 3 kw taps; weights are broadcast
//            LDS reads.
//   wgrad:   the same tile, but a workgroup owns ONE kd plane of the filter (9*Co accumulators per lane instead of 27*Co) and
//            walks its share of the tiles; the three kd workgroups of a share sit next to each other on one XCD so that dy is
//            fetched from HBM once.  Lanes are combined by wave shuffles + LDS in a fixed order, one partial per
//            workgroup, then wgrad_reduce_kernel (deterministic).
constexpr int C1D = 4, C1H = 8, C1WQ = 8;   // tile: 4 x 8 x (8*VPT) voxels, 256 lanes

template <int CO> struct Cin1 { static constexpr int VPT = CO <= 8 ? 4 : 2; };
typedef float f32x2 __attribute__((ext_vector_type(2)));



// stage rows [dlo, dlo+ND) x (C1H+2) x (TW+2) of the zero-padded input around tile origin (d0, h0, w0) into LDS
template <typename T, int TW, int ND, int NTHR = 256>
__device__ __forceinline__ void cin1_stage(float* xs, const T* __restrict__ xn, const Mri3dConvGeom& g, int dlo, int h0,
                                           int w0, int tid) {
    constexpr int PW = TW + 4, NE = ND * (C1H + 2) * (TW + 2);
    for (int e = tid; e < NE; e += NTHR) {
        const int fw = e % (TW + 2), r = e / (TW + 2);
        const int fh = r % (C1H + 2), fd = r / (C1H + 2);
        const int id = dlo + fd, ih = h0 - 1 + fh, iw = w0 - 1 + fw;
        const bool ok = (unsigned)id < (unsigned)g.di && (unsigned)ih < (unsigned)g.hi && (unsigned)iw < (unsigned)g.wi;
        const int cd = min(max(id, 0), g.di - 1), ch = min(max(ih, 0), g.hi - 1), cw = min(max(iw, 0), g.wi - 1);
        const float v = ldf(xn + (((int64_t)cd * g.hi + ch) * g.wi + cw) * g.x_ld);
        xs[(fd * (C1H + 2) + fh) * PW + fw] = ok ? v : 0.f;
    }
}

template <typename T, int CO>
__global__ void __launch_bounds__(256)
conv_cin1_fwd_kernel(Mri3dConvGeom g, const T* __restrict__ x, const float* __restrict__ wp, const float* __restrict__ bias,
                     T* __restrict__ y, int tilesD, int tilesH, int tilesW, int ntiles) {
    // wp = repacked weights [tap][co].  They live in LDS and are re-read (broadcast) per (kd, kh) row through a pointer the
    // compiler cannot see through: left to itself it hoists all 27*Co loop-invariant weights out of the tile loop
    // (398 VGPRs+AGPRs, one wave per SIMD; as scalar loads: spilled to VGPR lanes, 722 v_readlane per tile).
    constexpr int VPT = Cin1<CO>::VPT, TW = C1WQ * VPT, PW = TW + 4;
    __shared__ __attribute__((aligned(16))) float xs[(C1D + 2) * (C1H + 2) * PW];
    __shared__ __attribute__((aligned(16))) float wsh[27 * CO];
    const int tid = threadIdx.x;
    for (int e = tid; e < 27 * CO; e += 256) wsh[e] = wp[e];
    const int wq = tid % C1WQ, hl = (tid / C1WQ) % C1H, dl = tid / (C1WQ * C1H);
    float bv[CO];
#pragma unroll
    for (int c = 0; c < CO; ++c) bv[c] = bias ? bias[c] : 0.f;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int t = tile;
        const int w0 = (t % tilesW) * TW;
        t /= tilesW;
        const int h0 = (t % tilesH) * C1H;
        t /= tilesH;
        const int d0 = (t % tilesD) * C1D;
        const int n = t / tilesD;
        __syncthreads();
        cin1_stage<T, TW, C1D + 2>(xs, x + (int64_t)n * g.di * g.hi * g.wi * g.x_ld, g, d0 - 1, h0, w0, tid);
        __syncthreads();
        f32x2 acc[VPT][CO / 2];   // channel pairs: v_pk_fma_f32
#pragma unroll
        for (int v = 0; v < VPT; ++v)
#pragma unroll
            for (int c = 0; c < CO / 2; ++c) acc[v][c] = f32x2{bv[2 * c], bv[2 * c + 1]};
#pragma unroll
        for (int kd = 0; kd < 3; ++kd)
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                const float* row = xs + ((dl + kd) * (C1H + 2) + hl + kh) * PW + wq * VPT;
                float r[VPT + 2];
#pragma unroll
                for (int i = 0; i < VPT + 2; ++i) r[i] = row[i];
                int woff = (kd * 3 + kh) * 3 * CO;
                asm volatile("" : "+v"(woff));              // opaque offset: the weight reads stay inside the tile loop ...
                __builtin_amdgcn_sched_barrier(0);          // ... and inside their (kd, kh) group
                const float* wrow = wsh + woff;
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const float* wt = wrow + kw * CO;
#pragma unroll
                    for (int c = 0; c < CO / 2; ++c) {
                        const f32x2 wv = f32x2{wt[2 * c], wt[2 * c + 1]};
#pragma unroll
                        for (int v = 0; v < VPT; ++v)
                            acc[v][c] = __builtin_elementwise_fma(f32x2{r[v + kw], r[v + kw]}, wv, acc[v][c]);
                    }
                }
            }
        // results are pinned here: otherwise the compiler sinks each voxel's 27*Co FMAs into its bounds-checked store below
        // and keeps every weight and input row of the tile alive across all of them (256 VGPRs + AGPR traffic)
#pragma unroll
        for (int v = 0; v < VPT; ++v)
#pragma unroll
            for (int c = 0; c < CO / 2; ++c) asm volatile("" : "+v"(acc[v][c]));
        const int od = d0 + dl, oh = h0 + hl;
        if (od < g.dout && oh < g.ho) {
            T* yo = y + ((((int64_t)n * g.dout + od) * g.ho + oh) * g.wo + w0 + wq * VPT) * g.y_ld;
#pragma unroll
            for (int v = 0; v < VPT; ++v)
                if (w0 + wq * VPT + v < g.wo) {
#pragma unroll
                    for (int c = 0; c < CO; c += 4)
                        stf4(yo + (int64_t)v * g.y_ld + c,
                             make_float4(acc[v][c / 2].x, acc[v][c / 2].y, acc[v][c / 2 + 1].x, acc[v][c / 2 + 1].y));
                }
        }
    }
}

template <typename T, int CO>
__global__ void __launch_bounds__(768)
conv_cin1_wgrad_kernel(Mri3dConvGeom g, const T* __restrict__ x, const T* __restrict__ dy, float* __restrict__ part,
                       float* __restrict__ bias_part, int tilesD, int tilesH, int tilesW, int ntiles, int nshares) {
    constexpr int VPT = Cin1<CO>::VPT, TW = C1WQ * VPT, PW = TW + 4, NA = 9 * CO + CO;
    __shared__ __attribute__((aligned(16))) float xs[(C1D + 2) * (C1H + 2) * PW];
    __shared__ float red[12][NA];
    // One workgroup of 12 waves per share: waves 4 kd .. 4 kd + 3 take the taps of plane kd of the SAME tile — x is staged once for
    // the three of them and their dy loads, issued within microseconds of each other on one CU, are served from its L1 / L2 (as three
    // workgroups per share each of them fetched dy: 3 x the gradient tensor, 0.23 ms for 2 x 160x192x160 fp32).
    const int kd = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8);
    const int tid = threadIdx.x & 255;
    // blockIdx -> share, consecutive shares inside one XCD's contiguous range (gridDim.x = nshares, a multiple of 8)
    const int per_xcd = gridDim.x / 8;
    const int share = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
    const int wq = tid % C1WQ, hl = (tid / C1WQ) % C1H, dl = tid / (C1WQ * C1H);
    f32x2 acc[3][3][CO / 2];   // channel pairs: v_pk_fma_f32
    float bsum[CO];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
            for (int c = 0; c < CO / 2; ++c) acc[a][b][c] = f32x2{0.f, 0.f};
#pragma unroll
    for (int c = 0; c < CO; ++c) bsum[c] = 0.f;
    for (int tile = share; tile < ntiles; tile += nshares) {
        int t = tile;
        const int w0 = (t % tilesW) * TW;
        t /= tilesW;
        const int h0 = (t % tilesH) * C1H;
        t /= tilesH;
        const int d0 = (t % tilesD) * C1D;
        const int n = t / tilesD;
        __syncthreads();
        cin1_stage<T, TW, C1D + 2, 768>(xs, x + (int64_t)n * g.di * g.hi * g.wi * g.x_ld, g, d0 - 1, h0, w0, (int)threadIdx.x);
        // this lane's VPT output voxels of dy (zero outside the volume)
        const int od = d0 + dl, oh = h0 + hl;
        f32x2 gy[VPT][CO / 2];
#pragma unroll
        for (int v = 0; v < VPT; ++v) {
            const int ow = w0 + wq * VPT + v;
            const bool ok = od < g.dout && oh < g.ho && ow < g.wo;
            const T* dp = dy + ((((int64_t)n * g.dout + min(od, g.dout - 1)) * g.ho + min(oh, g.ho - 1)) * g.wo +
                                min(ow, g.wo - 1)) * g.y_ld;
#pragma unroll
            for (int c = 0; c < CO; c += 4) {
                const float4 q = ldf4(dp + c);
                gy[v][c / 2] = f32x2{ok ? q.x : 0.f, ok ? q.y : 0.f};
                gy[v][c / 2 + 1] = f32x2{ok ? q.z : 0.f, ok ? q.w : 0.f};
            }
        }
        __syncthreads();
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const float* row = xs + ((dl + kd) * (C1H + 2) + hl + kh) * PW + wq * VPT;
            float r[VPT + 2];
#pragma unroll
            for (int i = 0; i < VPT + 2; ++i) r[i] = row[i];
#pragma unroll
            for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                for (int v = 0; v < VPT; ++v)
#pragma unroll
                    for (int c = 0; c < CO / 2; ++c)
                        acc[kh][kw][c] = __builtin_elementwise_fma(f32x2{r[v + kw], r[v + kw]}, gy[v][c], acc[kh][kw][c]);
        }
        if (kd == 1) {
#pragma unroll
            for (int v = 0; v < VPT; ++v)
#pragma unroll
                for (int c = 0; c < CO / 2; ++c) { bsum[2 * c] += gy[v][c].x; bsum[2 * c + 1] += gy[v][c].y; }
        }
    }
    // lanes -> wave with DPP adds inside the 16-lane rows + the four row sums (fixed order), waves -> workgroup through LDS
    // (ds_bpermute shuffles here cost 480 LDS round trips per workgroup, 40 % of its run time)
    const int lane = tid & 63, wave = (int)threadIdx.x >> 6;   // 0 .. 11; waves 4 kd .. 4 kd + 3 hold plane kd's taps
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
            for (int c = 0; c < CO; ++c) {
                const float vsum = wave_sum_dpp((c & 1) ? acc[a][b][c / 2].y : acc[a][b][c / 2].x);
                if (lane == 0) red[wave][(a * 3 + b) * CO + c] = vsum;
            }
#pragma unroll
    for (int c = 0; c < CO; ++c) {
        const float vsum = wave_sum_dpp(bsum[c]);
        if (lane == 0) red[wave][9 * CO + c] = vsum;
    }
    __syncthreads();
    const int w0_ = 4 * kd;
    if (tid < 9 * CO) {
        const float t4 = (red[w0_][tid] + red[w0_ + 1][tid]) + (red[w0_ + 2][tid] + red[w0_ + 3][tid]);
        part[((size_t)share * 27 + kd * 9 + tid / CO) * CO + tid % CO] = t4;   // part[share][tap][ci = 0][co]
    } else if (tid < NA && kd == 1 && bias_part) {
        const int c = tid - 9 * CO;
        bias_part[(size_t)share * CO + c] = (red[w0_][tid] + red[w0_ + 1][tid]) + (red[w0_ + 2][tid] + red[w0_ + 3][tid]);
    }
}

// ------------------------------------------------------------------ 3x3x3 stencil: ONE input and ONE output channel
// The autoencoder's last layer `vox = Conv3d(1, 1, 3, padding=1)` (AE_model.py:160) on the whole reconstructed volume: 8 bytes of
// traffic and 27 FMAs per voxel.  The generic kernels spend 0.55 / 0.82 / 0.72 ms (fwd / dgrad / wgrad, 4 x 160x192x160) on it,
// 200-300 GB/s.  Same LDS tile as the first-layer kernels above; the 27 weights sit in SGPRs; dgrad = the same stencil with the
// taps reversed, applied to dy; wgrad keeps 27 accumulators per lane.
template <typename T>
__global__ void __launch_bounds__(256)
conv_c1c1_stencil_kernel(Mri3dConvGeom g, const T* __restrict__ x, int x_ld, const float* __restrict__ w, const float* __restrict__ bias,
                         int flip, T* __restrict__ y, int y_ld, int tilesD, int tilesH, int tilesW, int ntiles) {
    constexpr int VPT = 4, TW = C1WQ * VPT, PW = TW + 4;
    __shared__ __attribute__((aligned(16))) float xs[(C1D + 2) * (C1H + 2) * PW];
    const int tid = threadIdx.x;
    const int wq = tid % C1WQ, hl = (tid / C1WQ) % C1H, dl = tid / (C1WQ * C1H);
    float wt[27];
#pragma unroll
    for (int t = 0; t < 27; ++t) wt[t] = w[flip ? 26 - t : t];   // uniform -> scalar registers
    const float b0 = bias ? bias[0] : 0.f;
    Mri3dConvGeom gs = g;
    gs.x_ld = x_ld;   // cin1_stage reads its source pitch from the geometry
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int t = tile;
        const int w0 = (t % tilesW) * TW;
        t /= tilesW;
        const int h0 = (t % tilesH) * C1H;
        t /= tilesH;
        const int d0 = (t % tilesD) * C1D;
        const int n = t / tilesD;
        __syncthreads();
        cin1_stage<T, TW, C1D + 2>(xs, x + (int64_t)n * g.di * g.hi * g.wi * x_ld, gs, d0 - 1, h0, w0, tid);
        __syncthreads();
        float acc[VPT];
#pragma unroll
        for (int v = 0; v < VPT; ++v) acc[v] = b0;
#pragma unroll
        for (int kd = 0; kd < 3; ++kd)
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                const float* row = xs + ((dl + kd) * (C1H + 2) + hl + kh) * PW + wq * VPT;
                float r[VPT + 2];
#pragma unroll
                for (int i = 0; i < VPT + 2; ++i) r[i] = row[i];
#pragma unroll
                for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                    for (int v = 0; v < VPT; ++v) acc[v] = fmaf(r[v + kw], wt[(kd * 3 + kh) * 3 + kw], acc[v]);
            }
#pragma unroll
        for (int v = 0; v < VPT; ++v) asm volatile("" : "+v"(acc[v]));
        const int od = d0 + dl, oh = h0 + hl;
        if (od < g.di && oh < g.hi) {
            T* yo = y + ((((int64_t)n * g.di + od) * g.hi + oh) * g.wi + w0 + wq * VPT) * y_ld;
#pragma unroll
            for (int v = 0; v < VPT; ++v)
                if (w0 + wq * VPT + v < g.wi) stf(yo + (int64_t)v * y_ld, acc[v]);
        }
    }
}

template <typename T>
__global__ void __launch_bounds__(256)
conv_c1c1_wgrad_kernel(Mri3dConvGeom g, const T* __restrict__ x, const T* __restrict__ dy, float* __restrict__ part,
                       float* __restrict__ bias_part, int tilesD, int tilesH, int tilesW, int ntiles) {
    constexpr int VPT = 4, TW = C1WQ * VPT, PW = TW + 4;
    __shared__ __attribute__((aligned(16))) float xs[(C1D + 2) * (C1H + 2) * PW];
    __shared__ float red[4][28];
    const int tid = threadIdx.x;
    const int wq = tid % C1WQ, hl = (tid / C1WQ) % C1H, dl = tid / (C1WQ * C1H);
    float acc[27], bsum = 0.f;
#pragma unroll
    for (int t = 0; t < 27; ++t) acc[t] = 0.f;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int t = tile;
        const int w0 = (t % tilesW) * TW;
        t /= tilesW;
        const int h0 = (t % tilesH) * C1H;
        t /= tilesH;
        const int d0 = (t % tilesD) * C1D;
        const int n = t / tilesD;
        __syncthreads();
        cin1_stage<T, TW, C1D + 2>(xs, x + (int64_t)n * g.di * g.hi * g.wi * g.x_ld, g, d0 - 1, h0, w0, tid);
        const int od = d0 + dl, oh = h0 + hl;
        float gy[VPT];
#pragma unroll
        for (int v = 0; v < VPT; ++v) {
            const int ow = w0 + wq * VPT + v;
            const bool ok = od < g.dout && oh < g.ho && ow < g.wo;
            const float q = ldf(dy + ((((int64_t)n * g.dout + min(od, g.dout - 1)) * g.ho + min(oh, g.ho - 1)) * g.wo +
                                      min(ow, g.wo - 1)) * g.y_ld);
            gy[v] = ok ? q : 0.f;
            bsum += gy[v];
        }
        __syncthreads();
#pragma unroll
        for (int kd = 0; kd < 3; ++kd)
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                const float* row = xs + ((dl + kd) * (C1H + 2) + hl + kh) * PW + wq * VPT;
                float r[VPT + 2];
#pragma unroll
                for (int i = 0; i < VPT + 2; ++i) r[i] = row[i];
#pragma unroll
                for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                    for (int v = 0; v < VPT; ++v) acc[(kd * 3 + kh) * 3 + kw] = fmaf(r[v + kw], gy[v], acc[(kd * 3 + kh) * 3 + kw]);
            }
    }
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int t = 0; t < 27; ++t) {
        const float vsum = wave_sum_dpp(acc[t]);
        if (lane == 0) red[wave][t] = vsum;
    }
    {
        const float vsum = wave_sum_dpp(bsum);
        if (lane == 0) red[wave][27] = vsum;
    }
    __syncthreads();
    if (tid < 27) part[(size_t)blockIdx.x * 27 + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
    else if (tid == 27 && bias_part) bias_part[blockIdx.x] = (red[0][27] + red[1][27]) + (red[2][27] + red[3][27]);
}

constexpr int kC1C1Blocks = 1024;

// ------------------------------------------------------------------ 1 -> 1 channel, few taps (the autoencoder's last separable convs)
// Conv3d(1, 1, (1,3,1) / (1,1,3) ...) at full resolution (AE_model.py:110-160): a one-channel tensor is a dense float array, so a lane
// takes FOUR consecutive voxels of a row as one 16-byte access and a tap is the same access shifted by a wave-uniform element
// offset (4-byte aligned: global dwordx4 loads do not need more).  Forward and stride-1 data gradient are the same stencil (the
// gradient walks the taps with the opposite sign); the weight gradient is taps + 1 dot products reduced per wave (DPP), per
// workgroup (LDS, double) and across workgroups (wgrad_reduce_kernel: fixed order).  fp32, stride 1, <= 8 taps, W % 4 == 0.
// The gather kernels spent 0.11 - 0.18 ms per pass on 4 x 160x192x160 (1 TB/s) on 64-bit index arithmetic per voxel.
constexpr int kC1TBlocks = 1024;

// sign = +1: y[o] = b + sum_t w[t] x[o - p + k_t d]   (forward);   sign = -1: dx[i] = sum_t w[t] dy[i + p - k_t d]   (data gradient)
template <int NT>   // tap slots: 3 (the autoencoder's (3,1,1) / (1,3,1) / (1,1,3) filters) or kSmTaps
__global__ void __launch_bounds__(256)
conv_c1_taps_kernel(Mri3dConvGeom g, const float* __restrict__ src, const float* __restrict__ w, const float* __restrict__ bias,
                    float* __restrict__ dst, int sign, int sD, int sH, int sW, int oD, int oH, int oW) {
    const int taps = g.kd * g.kh * g.kw;
    const unsigned W4 = (unsigned)oW >> 2;
    const unsigned nitems = (unsigned)g.n * oD * oH * W4;   // < 2^31 (c1_taps_ok): 32-bit index arithmetic per item
    const float b0 = bias != nullptr ? bias[0] : 0.f;
    for (unsigned item = blockIdx.x * blockDim.x + threadIdx.x; item < nitems; item += gridDim.x * blockDim.x) {
        const unsigned q = item % W4;
        unsigned r = item / W4;
        const int oh = (int)(r % (unsigned)oH);
        r /= (unsigned)oH;
        const int od = (int)(r % (unsigned)oD), n = (int)(r / (unsigned)oD);
        const int ow = 4 * (int)q;
        const float* sn = src + (int64_t)n * sD * sH * sW;
        float4 xv[NT];
        unsigned vm[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int tt = t < taps ? t : 0;
            const int kw = tt % g.kw, kh = (tt / g.kw) % g.kh, kd = tt / (g.kw * g.kh);
            const int sd_ = od + sign * (kd * g.dd - g.pd), sh_ = oh + sign * (kh * g.dh - g.ph), sw_ = ow + sign * (kw * g.dw - g.pw);
            const bool rowok = t < taps && (unsigned)sd_ < (unsigned)sD && (unsigned)sh_ < (unsigned)sH;
            unsigned m = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) m |= (rowok && (unsigned)(sw_ + j) < (unsigned)sW) ? (1u << j) : 0u;
            vm[t] = m;
            const float* p = sn + ((int64_t)(rowok ? sd_ : 0) * sH + (rowok ? sh_ : 0)) * sW;
            if (m == 15u) {
                xv[t] = *reinterpret_cast<const float4*>(p + sw_);          // 4-byte aligned 16-byte load
            } else {   // row ends / outside: element-wise (a handful of lanes per row)
                xv[t].x = (m & 1u) ? p[sw_] : 0.f;
                xv[t].y = (m & 2u) ? p[sw_ + 1] : 0.f;
                xv[t].z = (m & 4u) ? p[sw_ + 2] : 0.f;
                xv[t].w = (m & 8u) ? p[sw_ + 3] : 0.f;
            }
        }
        float4 acc = make_float4(b0, b0, b0, b0);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (t < taps) {
                const float wt = w[t];
                acc.x = fmaf(xv[t].x, wt, acc.x);
                acc.y = fmaf(xv[t].y, wt, acc.y);
                acc.z = fmaf(xv[t].z, wt, acc.z);
                acc.w = fmaf(xv[t].w, wt, acc.w);
            }
        }
        *reinterpret_cast<float4*>(dst + (((int64_t)n * oD + od) * oH + oh) * oW + ow) = acc;
    }
}

// part[b][t] = sum over the workgroup's voxels of x[o - p + k_t d] * dy[o];  bias_part[b] = sum dy
template <int NT>
__global__ void __launch_bounds__(256)
conv_c1_taps_wgrad_kernel(Mri3dConvGeom g, const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ part,
                          float* __restrict__ bias_part) {
    __shared__ double red[4][NT + 1];
    const int taps = g.kd * g.kh * g.kw;
    const unsigned W4 = (unsigned)g.wo >> 2;
    const unsigned nitems = (unsigned)g.n * g.dout * g.ho * W4;   // < 2^31 (c1_taps_ok)
    float acc[NT + 1];
#pragma unroll
    for (int t = 0; t <= NT; ++t) acc[t] = 0.f;
    for (unsigned item = blockIdx.x * blockDim.x + threadIdx.x; item < nitems; item += gridDim.x * blockDim.x) {
        const unsigned q = item % W4;
        unsigned r = item / W4;
        const int oh = (int)(r % (unsigned)g.ho);
        r /= (unsigned)g.ho;
        const int od = (int)(r % (unsigned)g.dout), n = (int)(r / (unsigned)g.dout);
        const int ow = 4 * (int)q;
        const float4 gv = *reinterpret_cast<const float4*>(dy + (((int64_t)n * g.dout + od) * g.ho + oh) * g.wo + ow);
        acc[NT] += (gv.x + gv.y) + (gv.z + gv.w);
        const float* xn = x + (int64_t)n * g.di * g.hi * g.wi;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int tt = t < taps ? t : 0;
            const int kw = tt % g.kw, kh = (tt / g.kw) % g.kh, kd = tt / (g.kw * g.kh);
            const int id = od - g.pd + kd * g.dd, ih = oh - g.ph + kh * g.dh, iw = ow - g.pw + kw * g.dw;
            const bool rowok = t < taps && (unsigned)id < (unsigned)g.di && (unsigned)ih < (unsigned)g.hi;
            const float* p = xn + ((int64_t)(rowok ? id : 0) * g.hi + (rowok ? ih : 0)) * g.wi;
            float4 xv;
            if (rowok && iw >= 0 && iw + 3 < g.wi) {
                xv = *reinterpret_cast<const float4*>(p + iw);
            } else {
                xv.x = (rowok && (unsigned)iw < (unsigned)g.wi) ? p[iw] : 0.f;
                xv.y = (rowok && (unsigned)(iw + 1) < (unsigned)g.wi) ? p[iw + 1] : 0.f;
                xv.z = (rowok && (unsigned)(iw + 2) < (unsigned)g.wi) ? p[iw + 2] : 0.f;
                xv.w = (rowok && (unsigned)(iw + 3) < (unsigned)g.wi) ? p[iw + 3] : 0.f;
            }
            acc[t] = fmaf(xv.x, gv.x, fmaf(xv.y, gv.y, fmaf(xv.z, gv.z, fmaf(xv.w, gv.w, acc[t]))));
        }
    }
    // wave sums (fixed butterfly), then the four waves in double through LDS
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int t = 0; t <= NT; ++t) {
        float v = acc[t];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane == 0) red[wv][t] = (double)v;
    }
    __syncthreads();
    if (threadIdx.x <= NT) {
        const double sdbl = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
        if ((int)threadIdx.x < taps) part[(size_t)blockIdx.x * taps + threadIdx.x] = (float)sdbl;
        if (threadIdx.x == NT && bias_part != nullptr) bias_part[blockIdx.x] = (float)sdbl;
    }
}

static bool c1_taps_ok(const Mri3dConvGeom& g) {
    return g.dtype == MRI3D_F32 && g.ci == 1 && g.co == 1 && g.kd * g.kh * g.kw <= kSmTaps && g.sd == 1 && g.sh == 1 && g.sw == 1 &&
           g.x_ld == 1 && g.y_ld == 1 && g.wo % 4 == 0 && g.wi % 4 == 0 && !(g.kd == 3 && g.kh == 3 && g.kw == 3) &&
           (int64_t)g.n * g.dout * g.ho * g.wo < 0x7fffffffLL && (int64_t)g.n * g.di * g.hi * g.wi < 0x7fffffffLL;
}

static bool c1c1_ok(const Mri3dConvGeom& g) {
    static const int off = tuning_knob("MRI3D_C1C1_OFF", 0);   // tuning aid (A/B)
    return !off && g.ci == 1 && g.co == 1 && g.kd == 3 && g.kh == 3 && g.kw == 3 && g.sd == 1 && g.sh == 1 && g.sw == 1 &&
           g.dd == 1 && g.dh == 1 && g.dw == 1 && g.pd == 1 && g.ph == 1 && g.pw == 1 &&
           (int64_t)g.n * cdiv(g.dout, C1D) * cdiv(g.ho, C1H) * cdiv(g.wo, C1WQ * 4) < 0x7fffffffLL;
}

template <typename T>
static void launch_c1c1_stencil(const Mri3dConvGeom& g, const T* src, int src_ld, const float* w, const float* bias, int flip,
                                T* dst, int dst_ld, hipStream_t s) {
    const int tilesD = cdiv(g.di, C1D), tilesH = cdiv(g.hi, C1H), tilesW = cdiv(g.wi, C1WQ * 4);
    const int ntiles = g.n * tilesD * tilesH * tilesW;
    hipLaunchKernelGGL(conv_c1c1_stencil_kernel<T>, dim3(std::min(ntiles, 256 * 8)), dim3(256), 0, s, g, src, src_ld, w, bias, flip,
                       dst, dst_ld, tilesD, tilesH, tilesW, ntiles);
}

static const int g_cin1_off = tuning_knob("MRI3D_CIN1_OFF", 0);   // tuning aid (A/B)
constexpr int kCin1Shares = 256;   // 3 * 256 = 768 workgroups = 3 per CU

static bool cin1_ok(const Mri3dConvGeom& g) {
    return !g_cin1_off && g.ci == 1 && (g.co == 8 || g.co == 16) && g.kd == 3 && g.kh == 3 && g.kw == 3 && g.sd == 1 &&
           g.sh == 1 && g.sw == 1 && g.dd == 1 && g.dh == 1 && g.dw == 1 && g.pd == 1 && g.ph == 1 && g.pw == 1 &&
           g.y_ld % 4 == 0 && (int64_t)g.n * cdiv(g.dout, C1D) * cdiv(g.ho, C1H) * cdiv(g.wo, C1WQ * 2) < 0x7fffffffLL;
}

static bool wgrad_small_ok(const Mri3dConvGeom& g) {
    const int taps = g.kd * g.kh * g.kw;
    return taps <= kSmTaps && g.ci <= 64 && g.co <= 64 && (int64_t)g.dout * g.ho * g.wo < 0x7fffffffLL;
}

// the channel-quad variant: Ci, Co multiples of 4 with QI * QO lanes per voxel dividing 256 (<= 64), 16-byte aligned rows
static bool wgrad_quads_ok(const Mri3dConvGeom& g) {
    const int taps = g.kd * g.kh * g.kw;
    const int pv = (g.ci / 4) * (g.co / 4);
    if (g.ci == 1)   // one input channel: a lane owns (voxel, co quad)
        return taps <= kSmTaps && g.co % 4 == 0 && g.co >= 4 && g.co <= 64 && 256 % (g.co / 4) == 0 && g.y_ld % 4 == 0 &&
               (int64_t)g.ho * g.wo < 0x7fffffffLL && (int64_t)g.n * g.dout * g.ho < 0x7fffffffLL && !cin1_ok(g);
    return taps <= kSmTaps && g.ci % 4 == 0 && g.co % 4 == 0 && g.ci >= 4 && g.co >= 4 && pv <= 64 && 256 % pv == 0 &&
           g.x_ld % 4 == 0 && g.y_ld % 4 == 0 && (int64_t)g.ho * g.wo < 0x7fffffffLL &&
           (int64_t)g.n * g.dout * g.ho < 0x7fffffffLL;
}
struct WgradQuadsPlan { int QI, QO, hch, gx, CiP, CoP; size_t part_floats, bias_floats; };
static WgradQuadsPlan wgrad_quads_plan(const Mri3dConvGeom& g) {
    WgradQuadsPlan p;
    p.QI = g.ci == 1 ? 1 : g.ci / 4, p.QO = g.co / 4, p.CiP = g.ci, p.CoP = g.co;
    const int VL = 256 / (p.QI * p.QO);
    // ~16 voxels per lane and slab
    p.hch = (int)std::max<int64_t>(1, std::min<int64_t>(g.ho, (int64_t)16 * VL / std::max(g.wo, 1)));
    const int64_t slabs = (int64_t)g.n * g.dout * cdiv(g.ho, p.hch);
    // one resident round: 4 / 3 / 2 workgroups per CU with 4 / 6 / 8 accumulator slots (112 / 161 / 209 VGPRs)
    const int taps = g.kd * g.kh * g.kw;
    const int resident = 256 * (taps <= 4 ? 4 : (taps <= 6 ? 3 : 2));
    p.gx = (int)std::max<int64_t>(1, std::min<int64_t>(slabs, resident));
    p.part_floats = (size_t)p.gx * g.kd * g.kh * g.kw * p.CiP * p.CoP;
    p.bias_floats = (size_t)p.gx * g.co;
    return p;
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
    if (cin1_ok(g)) a = std::max(a, (size_t)kCin1Shares * (27 + 1) * g.co * sizeof(float));
    if (c1c1_ok(g)) a = std::max(a, (size_t)kC1C1Blocks * 28 * sizeof(float));
    if (c1_taps_ok(g)) a = std::max(a, (size_t)kC1TBlocks * (kSmTaps + 1) * sizeof(float));
    if (wgrad_co1_ok(g)) a = std::max(a, (size_t)kCo1Blocks * (kSmTaps * 16 + 1) * sizeof(float));
    if (wgrad_small_ok(g)) {
        WgradSmallPlan q = wgrad_small_plan(g);
        a = std::max(a, (q.part_floats + q.bias_floats) * sizeof(float));
    }
    if (wgrad_quads_ok(g)) {
        WgradQuadsPlan q = wgrad_quads_plan(g);
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
    const int taps = g.kd * g.kh * g.kw;
    if ((vec || g.ci == 1) && taps <= 8 && aligned_vec4(g.dtype, y) && (int64_t)g.ho * g.wo < 0x7fffffffLL && (int64_t)g.n * g.dout * g.ho < 0x7fffffffLL) {
        // few taps: slab walk + batched tap loads (conv_fwd_taps_kernel)
        const int hch = (int)std::max<int64_t>(1, std::min<int64_t>(g.ho, (int64_t)2048 / std::max(g.wo, 1)));
        const int64_t slabs = (int64_t)g.n * g.dout * cdiv(g.ho, hch);
        dim3 tgrid((unsigned)std::min<int64_t>(slabs, 4096), CoP / TL);
#define MRI3D_FT(NTv, CVv) hipLaunchKernelGGL((conv_fwd_taps_kernel<T, TL, NTv, CVv>), tgrid, dim3(256), 0, s, g, (const T*)x, wp, bias, (T*)y, CoP, hch)
        MRI3D_DISPATCH_DTYPE(g.dtype, T, {
            // tap slots = the filter's taps where an instance exists (3: the decoder's k = 3 filters, 6: the encoder's k = 6 ones): a
            // slot past the filter still costs its address arithmetic and a (cached) load
            if (vec) { if (taps <= 3) MRI3D_FT(3, 4); else if (taps <= 4) MRI3D_FT(4, 4); else if (taps <= 6) MRI3D_FT(6, 4); else MRI3D_FT(8, 4); }
            else { if (taps <= 3) MRI3D_FT(3, 1); else if (taps <= 6) MRI3D_FT(6, 1); else MRI3D_FT(8, 1); }
        });
#undef MRI3D_FT
        return;
    }
    MRI3D_DISPATCH_DTYPE(g.dtype, T, {
        if (vec)
            hipLaunchKernelGGL((conv_fwd_generic_kernel<T, TL, true>), grid, dim3(256), 0, s, g, (const T*)x, wp, bias, (T*)y, CoP);
        else
            hipLaunchKernelGGL((conv_fwd_generic_kernel<T, TL, false>), grid, dim3(256), 0, s, g, (const T*)x, wp, bias, (T*)y, CoP);
    });
}

int conv_generic_fwd(const Mri3dConvGeom& g, const void* x, const float* w, const float* bias, void* y, void* ws,
                     size_t ws_bytes, hipStream_t s) {
    if (c1c1_ok(g)) {
        MRI3D_DISPATCH_DTYPE(g.dtype, T, { launch_c1c1_stencil<T>(g, (const T*)x, g.x_ld, w, bias, 0, (T*)y, g.y_ld, s); });
        return check_launch("conv3d_fwd(1->1 stencil)");
    }
    if (c1_taps_ok(g) && aligned16(y)) {
        const int64_t items = (int64_t)g.n * g.dout * g.ho * (g.wo / 4);
        const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(cdiv64(items, 256), 8192));
        if (g.kd * g.kh * g.kw <= 3)
            hipLaunchKernelGGL(conv_c1_taps_kernel<3>, dim3(grid), dim3(256), 0, s, g, (const float*)x, w, bias, (float*)y, 1, g.di, g.hi,
                               g.wi, g.dout, g.ho, g.wo);
        else
            hipLaunchKernelGGL(conv_c1_taps_kernel<kSmTaps>, dim3(grid), dim3(256), 0, s, g, (const float*)x, w, bias, (float*)y, 1, g.di,
                               g.hi, g.wi, g.dout, g.ho, g.wo);
        return check_launch("conv3d_fwd(1->1 taps)");
    }
    if (cin1_ok(g) && aligned_vec4(g.dtype, y)) {
        const size_t need1 = (size_t)27 * g.co * sizeof(float);
        MRI3D_REQUIRE(ws != nullptr && ws_bytes >= need1, MRI3D_EWORKSPACE, "conv3d_fwd: workspace %zu < %zu", ws_bytes, need1);
        float* wp = static_cast<float*>(ws);
        hipLaunchKernelGGL(repack_w_fwd_kernel, dim3(1), dim3(256), 0, s, w, wp, g.co, 1, 27, g.co);
        const int vpt = g.co <= 8 ? 4 : 2;
        const int tilesD = cdiv(g.dout, C1D), tilesH = cdiv(g.ho, C1H), tilesW = cdiv(g.wo, C1WQ * vpt);
        const int ntiles = g.n * tilesD * tilesH * tilesW;
        const int grid = std::min(ntiles, 256 * 8);
        MRI3D_DISPATCH_DTYPE(g.dtype, T, {
            if (g.co == 8)
                hipLaunchKernelGGL((conv_cin1_fwd_kernel<T, 8>), dim3(grid), dim3(256), 0, s, g, (const T*)x, wp, bias, (T*)y,
                                   tilesD, tilesH, tilesW, ntiles);
            else
                hipLaunchKernelGGL((conv_cin1_fwd_kernel<T, 16>), dim3(grid), dim3(256), 0, s, g, (const T*)x, wp, bias, (T*)y,
                                   tilesD, tilesH, tilesW, ntiles);
        });
        return check_launch("conv3d_fwd(cin1)");
    }
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

// Data gradient of a STRIDED convolution (dilation 1): only the taps with k = (i + pad) mod stride reach an input voxel i, i.e.
// 1 .. 8 of the 27 taps of a 3x3x3 / stride-2 layer (modified_3dunet.py:23-38, cnn_model.py:49-81).  The gather kernel above
// walks all taps and masks (8x wasted FMAs: 6 TFLOP/s on the 8 -> 16 stride-2 layer at 80x96x80).  Here one WAVE owns one
// (n, id, ih, w-parity) row segment: the valid tap set is wave-uniform, so the tap loops just step by the stride, the weights
// stay on the scalar path and nothing is masked except the volume border.
template <typename T, int CIT, bool VEC4>
__global__ void __launch_bounds__(64)
conv_dgrad_strided_kernel(Mri3dConvGeom g, const T* __restrict__ dy, const float* __restrict__ wp,
                          const float* __restrict__ bias, T* __restrict__ dx, int CiP) {
    int u = blockIdx.x;
    const int rw = u % g.sw;   // residue of iw modulo the W stride handled by this wave
    u /= g.sw;
    const int ih = u % g.hi;
    u /= g.hi;
    const int id = u % g.di;
    const int n = u / g.di;
    const int cit = blockIdx.y * CIT;
    const int kd0 = (id + g.pd) % g.sd, kh0 = (ih + g.ph) % g.sh, kw0 = (rw + g.pw) % g.sw;
    for (int iw = rw + (int)threadIdx.x * g.sw; iw < g.wi; iw += 64 * g.sw) {
        float acc[CIT];
#pragma unroll
        for (int j = 0; j < CIT; ++j) acc[j] = (bias != nullptr && cit + j < g.ci) ? bias[cit + j] : 0.f;
        for (int kd = kd0; kd < g.kd; kd += g.sd) {
            const int nd = id + g.pd - kd;
            if (nd < 0) break;
            const int od = nd / g.sd;
            if (od >= g.dout) continue;
            for (int kh = kh0; kh < g.kh; kh += g.sh) {
                const int nh = ih + g.ph - kh;
                if (nh < 0) break;
                const int oh = nh / g.sh;
                if (oh >= g.ho) continue;
                for (int kw = kw0; kw < g.kw; kw += g.sw) {
                    const int nw = iw + g.pw - kw;
                    const int ow = nw / g.sw;
                    const bool valid = nw >= 0 && ow < g.wo;
                    const T* yp = dy + ((((int64_t)n * g.dout + od) * g.ho + oh) * g.wo + (valid ? ow : 0)) * g.y_ld;
                    const float* wt = wp + (size_t)((kd * g.kh + kh) * g.kw + kw) * g.co * CiP + cit;
                    if (VEC4) {
                        for (int co = 0; co < g.co; co += 4) {
                            const float4 gv = valid ? ldf4(yp + co) : make_float4(0.f, 0.f, 0.f, 0.f);
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
                            const float gv = valid ? ldf(yp + co) : 0.f;
                            const float* w0 = wt + (size_t)co * CiP;
#pragma unroll
                            for (int j = 0; j < CIT; ++j) acc[j] = fmaf(gv, w0[j], acc[j]);
                        }
                    }
                }
            }
        }
        T* xp = dx + ((((int64_t)n * g.di + id) * g.hi + ih) * g.wi + iw) * g.x_ld + cit;
#pragma unroll
        for (int j = 0; j < CIT; ++j)
            if (cit + j < g.ci) stf(xp + j, acc[j]);
    }
}

// The same wave-per-(n, id, ih, w-residue) decomposition with the valid taps ENUMERATED first (they are wave-uniform: at most NT of
// them) and, per output-channel quad, their dy loads issued together before the FMAs — in the kernel above every tap's load sits in
// a run-time loop behind `valid ? load : 0` and gets its own wait (6 serial round trips per voxel of the (1,6,1) stride-2 layer).
// Slots past the last valid tap re-read slot 0 (cached) and are masked: no branch around a load.
template <typename T, int CIT, int NT>
__global__ void __launch_bounds__(64)
conv_dgrad_strided_taps_kernel(Mri3dConvGeom g, const T* __restrict__ dy, const float* __restrict__ wp,
                               const float* __restrict__ bias, T* __restrict__ dx, int CiP) {
    int u = blockIdx.x;
    const int rw = u % g.sw;
    u /= g.sw;
    const int ih = u % g.hi;
    u /= g.hi;
    const int id = u % g.di;
    const int n = u / g.di;
    const int cit = blockIdx.y * CIT;
    const int kd0 = (id + g.pd) % g.sd, kh0 = (ih + g.ph) % g.sh, kw0 = (rw + g.pw) % g.sw;
    // enumerate the taps that reach this row (uniform): dy row base, kw, packed-weight tap index
    int64_t rowoff[NT];
    int tkw[NT], ttap[NT];
    int ntap = 0;
#pragma unroll
    for (int t = 0; t < NT; ++t) { rowoff[t] = 0; tkw[t] = kw0; ttap[t] = 0; }
    for (int kd = kd0; kd < g.kd; kd += g.sd) {
        const int nd = id + g.pd - kd;
        if (nd < 0) break;
        const int od = nd / g.sd;
        if (od >= g.dout) continue;
        for (int kh = kh0; kh < g.kh; kh += g.sh) {
            const int nh = ih + g.ph - kh;
            if (nh < 0) break;
            const int oh = nh / g.sh;
            if (oh >= g.ho) continue;
            for (int kw = kw0; kw < g.kw; kw += g.sw) {
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    if (t == ntap) {
                        rowoff[t] = ((((int64_t)n * g.dout + od) * g.ho + oh) * g.wo) * g.y_ld;
                        tkw[t] = kw;
                        ttap[t] = (kd * g.kh + kh) * g.kw + kw;
                    }
                ++ntap;   // host guarantees <= NT
            }
        }
    }
    // input voxel iw = rw + k * sw: its tap kw (kw = (rw + pw) mod sw, enumerated above) reaches output voxel k + (rw + pw - kw) / sw —
    // an exact, wave-uniform quotient per tap, so the voxel loop has no integer division
    int tq[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) tq[t] = (rw + g.pw - tkw[t]) / g.sw;
    int k = (int)threadIdx.x;
    for (int iw = rw + (int)threadIdx.x * g.sw; iw < g.wi; iw += 64 * g.sw, k += 64) {
        float acc[CIT];
#pragma unroll
        for (int j = 0; j < CIT; ++j) acc[j] = (bias != nullptr && cit + j < g.ci) ? bias[cit + j] : 0.f;
        int64_t off[NT];
        unsigned okm = 0;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int ow = k + tq[t];
            const bool ok = t < ntap && (unsigned)ow < (unsigned)g.wo;
            okm |= ok ? (1u << t) : 0u;
            off[t] = rowoff[t] + (int64_t)(ok ? ow : 0) * g.y_ld;
        }
        for (int co = 0; co < g.co; co += 4) {
            float4 gv[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) gv[t] = ldf4(dy + off[t] + co);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const bool ok = (okm >> t) & 1u;
                const float g0 = ok ? gv[t].x : 0.f, g1 = ok ? gv[t].y : 0.f, g2 = ok ? gv[t].z : 0.f, g3 = ok ? gv[t].w : 0.f;
                const float* w0 = wp + ((size_t)ttap[t] * g.co + co) * CiP + cit;
#pragma unroll
                for (int j = 0; j < CIT; ++j) {
                    acc[j] = fmaf(g0, w0[j], acc[j]);
                    acc[j] = fmaf(g1, w0[CiP + j], acc[j]);
                    acc[j] = fmaf(g2, w0[2 * CiP + j], acc[j]);
                    acc[j] = fmaf(g3, w0[3 * CiP + j], acc[j]);
                }
            }
        }
        T* xp = dx + ((((int64_t)n * g.di + id) * g.hi + ih) * g.wi + iw) * g.x_ld + cit;
        if (CIT % 4 == 0 && cit + CIT <= g.ci && (g.x_ld & 3) == 0) {
#pragma unroll
            for (int j = 0; j < CIT; j += 4) stf4(xp + j, make_float4(acc[j], acc[j + 1], acc[j + 2], acc[j + 3]));
        } else {
#pragma unroll
            for (int j = 0; j < CIT; ++j)
                if (cit + j < g.ci) stf(xp + j, acc[j]);
        }
    }
}

template <int TL>
static void launch_dgrad(const Mri3dConvGeom& g, const void* dy, const float* wp, const float* bias, void* dx,
                         int CiP, hipStream_t s) {
    int64_t nvox = (int64_t)g.n * g.di * g.hi * g.wi;
    dim3 grid((unsigned)std::min<int64_t>(cdiv64(nvox, 256), 8192), CiP / TL);
    bool vec = (g.co % 4 == 0) && (g.y_ld % 4 == 0) && aligned_vec4(g.dtype, dy);
    const int taps = g.kd * g.kh * g.kw;
    const bool unit_stride = g.sd == 1 && g.sh == 1 && g.sw == 1;   // strided layers: the wave-uniform tap sets below are faster (0.29 vs 0.33 ms)
    if ((vec || g.co == 1) && unit_stride && taps <= 8 && aligned_vec4(g.dtype, dx) && (int64_t)g.hi * g.wi < 0x7fffffffLL && (int64_t)g.n * g.di * g.hi < 0x7fffffffLL) {
        // few taps, stride 1: slab walk + batched tap loads (conv_dgrad_taps_kernel)
        const int hch = (int)std::max<int64_t>(1, std::min<int64_t>(g.hi, (int64_t)2048 / std::max(g.wi, 1)));
        const int64_t slabs = (int64_t)g.n * g.di * cdiv(g.hi, hch);
        dim3 tgrid((unsigned)std::min<int64_t>(slabs, 4096), CiP / TL);
#define MRI3D_DT(NTv, CVv) hipLaunchKernelGGL((conv_dgrad_taps_kernel<T, TL, NTv, CVv>), tgrid, dim3(256), 0, s, g, (const T*)dy, wp, bias, (T*)dx, CiP, hch)
        MRI3D_DISPATCH_DTYPE(g.dtype, T, {
            if (vec) { if (taps <= 3) MRI3D_DT(3, 4); else if (taps <= 4) MRI3D_DT(4, 4); else if (taps <= 6) MRI3D_DT(6, 4); else MRI3D_DT(8, 4); }
            else { if (taps <= 3) MRI3D_DT(3, 1); else MRI3D_DT(8, 1); }
        });
#undef MRI3D_DT
        return;
    }
    const int64_t rows = (int64_t)g.n * g.di * g.hi * g.sw;
    if ((g.sd > 1 || g.sh > 1 || g.sw > 1) && g.dd == 1 && g.dh == 1 && g.dw == 1 && rows <= 0x7fffffff) {
        dim3 sgrid((unsigned)rows, CiP / TL);
        // valid taps per input voxel: ceil(k / s) per axis; up to 4 (the separable stride-2 filters: 3) or 8 take the batched kernel
        const int vt = cdiv(g.kd, g.sd) * cdiv(g.kh, g.sh) * cdiv(g.kw, g.sw);
        if (vec && vt <= 8 && aligned_vec4(g.dtype, dx)) {
            MRI3D_DISPATCH_DTYPE(g.dtype, T, {
                if (vt <= 3)
                    hipLaunchKernelGGL((conv_dgrad_strided_taps_kernel<T, TL, 3>), sgrid, dim3(64), 0, s, g, (const T*)dy, wp, bias, (T*)dx, CiP);
                else if (vt <= 4)
                    hipLaunchKernelGGL((conv_dgrad_strided_taps_kernel<T, TL, 4>), sgrid, dim3(64), 0, s, g, (const T*)dy, wp, bias, (T*)dx, CiP);
                else
                    hipLaunchKernelGGL((conv_dgrad_strided_taps_kernel<T, TL, 8>), sgrid, dim3(64), 0, s, g, (const T*)dy, wp, bias, (T*)dx, CiP);
            });
            return;
        }
        MRI3D_DISPATCH_DTYPE(g.dtype, T, {
            if (vec)
                hipLaunchKernelGGL((conv_dgrad_strided_kernel<T, TL, true>), sgrid, dim3(64), 0, s, g, (const T*)dy, wp, bias, (T*)dx, CiP);
            else
                hipLaunchKernelGGL((conv_dgrad_strided_kernel<T, TL, false>), sgrid, dim3(64), 0, s, g, (const T*)dy, wp, bias, (T*)dx, CiP);
        });
        return;
    }
    MRI3D_DISPATCH_DTYPE(g.dtype, T, {
        if (vec)
            hipLaunchKernelGGL((conv_dgrad_generic_kernel<T, TL, true>), grid, dim3(256), 0, s, g, (const T*)dy, wp, bias, (T*)dx, CiP);
        else
            hipLaunchKernelGGL((conv_dgrad_generic_kernel<T, TL, false>), grid, dim3(256), 0, s, g, (const T*)dy, wp, bias, (T*)dx, CiP);
    });
}

int conv_generic_dgrad(const Mri3dConvGeom& g, const void* dy, const float* w, const float* bias, void* dx, void* ws,
                       size_t ws_bytes, hipStream_t s) {
    if (c1c1_ok(g)) {
        // dx = the same stencil with reversed taps, applied to dy (pad 1, stride 1: the transposed conv is a conv)
        MRI3D_DISPATCH_DTYPE(g.dtype, T, { launch_c1c1_stencil<T>(g, (const T*)dy, g.y_ld, w, bias, 1, (T*)dx, g.x_ld, s); });
        return check_launch("conv3d_dgrad(1->1 stencil)");
    }
    if (c1_taps_ok(g) && aligned16(dx)) {
        const int64_t items = (int64_t)g.n * g.di * g.hi * (g.wi / 4);
        const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(cdiv64(items, 256), 8192));
        if (g.kd * g.kh * g.kw <= 3)
            hipLaunchKernelGGL(conv_c1_taps_kernel<3>, dim3(grid), dim3(256), 0, s, g, (const float*)dy, w, bias, (float*)dx, -1, g.dout,
                           g.ho, g.wo, g.di, g.hi, g.wi);
        else
            hipLaunchKernelGGL(conv_c1_taps_kernel<kSmTaps>, dim3(grid), dim3(256), 0, s, g, (const float*)dy, w, bias, (float*)dx, -1, g.dout,
                           g.ho, g.wo, g.di, g.hi, g.wi);
        return check_launch("conv3d_dgrad(1->1 taps)");
    }
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
    if (c1c1_ok(g)) {
        const size_t need = (size_t)kC1C1Blocks * 28 * sizeof(float);
        MRI3D_REQUIRE(ws != nullptr && ws_bytes >= need, MRI3D_EWORKSPACE, "conv3d_wgrad: workspace %zu < %zu", ws_bytes, need);
        const int tilesD = cdiv(g.dout, C1D), tilesH = cdiv(g.ho, C1H), tilesW = cdiv(g.wo, C1WQ * 4);
        const int ntiles = g.n * tilesD * tilesH * tilesW;
        const int nb = std::min(ntiles, kC1C1Blocks);
        float* part = static_cast<float*>(ws);
        float* bias_part = dbias ? part + (size_t)kC1C1Blocks * 27 : nullptr;
        MRI3D_DISPATCH_DTYPE(g.dtype, T, {
            hipLaunchKernelGGL(conv_c1c1_wgrad_kernel<T>, dim3(nb), dim3(256), 0, s, g, (const T*)x, (const T*)dy, part, bias_part,
                               tilesD, tilesH, tilesW, ntiles);
        });
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(cdiv(27 + 1, 8)), dim3(256), 0, s, part, bias_part, dw, dbias, nb, 27, 1, 1, 1, 1);
        return check_launch("conv3d_wgrad(1->1 stencil)");
    }
    if (c1_taps_ok(g) && aligned16(dy)) {
        const int taps = g.kd * g.kh * g.kw;
        const int64_t items = (int64_t)g.n * g.dout * g.ho * (g.wo / 4);
        const int nb = (int)std::max<int64_t>(1, std::min<int64_t>(cdiv64(items, 256 * 4), kC1TBlocks));
        const size_t need = (size_t)kC1TBlocks * (kSmTaps + 1) * sizeof(float);
        MRI3D_REQUIRE(ws != nullptr && ws_bytes >= need, MRI3D_EWORKSPACE, "conv3d_wgrad: workspace %zu < %zu", ws_bytes, need);
        float* part = static_cast<float*>(ws);
        float* bias_part = dbias ? part + (size_t)kC1TBlocks * kSmTaps : nullptr;
        if (g.kd * g.kh * g.kw <= 3)
            hipLaunchKernelGGL(conv_c1_taps_wgrad_kernel<3>, dim3(nb), dim3(256), 0, s, g, (const float*)x, (const float*)dy, part, bias_part);
        else
            hipLaunchKernelGGL(conv_c1_taps_wgrad_kernel<kSmTaps>, dim3(nb), dim3(256), 0, s, g, (const float*)x, (const float*)dy, part, bias_part);
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(cdiv(taps + 1, 8)), dim3(256), 0, s, part, bias_part, dw, dbias, nb, taps, 1, 1, 1, 1);
        return check_launch("conv3d_wgrad(1->1 taps)");
    }
    if (cin1_ok(g) && aligned_vec4(g.dtype, dy)) {
        const size_t need = (size_t)kCin1Shares * (27 + 1) * g.co * sizeof(float);
        MRI3D_REQUIRE(ws != nullptr && ws_bytes >= need, MRI3D_EWORKSPACE, "conv3d_wgrad: workspace %zu < %zu", ws_bytes, need);
        const int vpt = g.co <= 8 ? 4 : 2;
        const int tilesD = cdiv(g.dout, C1D), tilesH = cdiv(g.ho, C1H), tilesW = cdiv(g.wo, C1WQ * vpt);
        const int ntiles = g.n * tilesD * tilesH * tilesW;
        float* part = static_cast<float*>(ws);
        float* bias_part = dbias ? part + (size_t)kCin1Shares * 27 * g.co : nullptr;
        // every (share, kd) workgroup writes its slots, also when its share of the tiles is empty (zeros)
        MRI3D_DISPATCH_DTYPE(g.dtype, T, {
            if (g.co == 8)
                hipLaunchKernelGGL((conv_cin1_wgrad_kernel<T, 8>), dim3(kCin1Shares), dim3(768), 0, s, g, (const T*)x,
                                   (const T*)dy, part, bias_part, tilesD, tilesH, tilesW, ntiles, kCin1Shares);
            else
                hipLaunchKernelGGL((conv_cin1_wgrad_kernel<T, 16>), dim3(kCin1Shares), dim3(768), 0, s, g, (const T*)x,
                                   (const T*)dy, part, bias_part, tilesD, tilesH, tilesW, ntiles, kCin1Shares);
        });
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(cdiv(27 * g.co + g.co, 8)), dim3(256), 0, s, part, bias_part, dw, dbias,
                           kCin1Shares, 27, 1, g.co, 1, g.co);
        return check_launch("conv3d_wgrad(cin1)");
    }
    if (wgrad_co1_ok(g) && aligned_vec4(g.dtype, x)) {
        const size_t need = (size_t)kCo1Blocks * (kSmTaps * 16 + 1) * sizeof(float);
        MRI3D_REQUIRE(ws != nullptr && ws_bytes >= need, MRI3D_EWORKSPACE, "conv3d_wgrad: workspace %zu < %zu", ws_bytes, need);
        const int taps = g.kd * g.kh * g.kw;
        const int64_t nvox = (int64_t)g.n * g.dout * g.ho * g.wo;
        const int nb = (int)std::min<int64_t>(cdiv64(nvox, 256), kCo1Blocks);
        float* part = static_cast<float*>(ws);
        float* bias_part = dbias ? part + (size_t)kCo1Blocks * kSmTaps * 16 : nullptr;
#define MRI3D_CO1(CIv) hipLaunchKernelGGL((conv_wgrad_co1_kernel<T, CIv>), dim3(nb), dim3(256), 0, s, g, (const T*)x, (const T*)dy, part, bias_part)
        MRI3D_DISPATCH_DTYPE(g.dtype, T, {
            switch (g.ci) {
                case 1: MRI3D_CO1(1); break;
                case 4: MRI3D_CO1(4); break;
                case 8: MRI3D_CO1(8); break;
                default: MRI3D_CO1(16); break;
            }
        });
#undef MRI3D_CO1
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(cdiv(taps * g.ci + 1, 8)), dim3(256), 0, s, part, bias_part, dw, dbias, nb, taps,
                           g.ci, 1, g.ci, 1);
        return check_launch("conv3d_wgrad(co1)");
    }
    if (wgrad_quads_ok(g) && aligned_vec4(g.dtype, dy) && (g.ci == 1 || aligned_vec4(g.dtype, x))) {
        WgradQuadsPlan q = wgrad_quads_plan(g);
        const size_t need = (q.part_floats + q.bias_floats) * sizeof(float);
        MRI3D_REQUIRE(ws != nullptr && ws_bytes >= need, MRI3D_EWORKSPACE, "conv3d_wgrad: workspace %zu < %zu", ws_bytes, need);
        float* part = static_cast<float*>(ws);
        float* bias_part = dbias ? part + q.part_floats : nullptr;
        const int taps = g.kd * g.kh * g.kw;
#define MRI3D_WGQ(NTv, CIVv)                                                                                           \
    hipLaunchKernelGGL((conv_wgrad_quads_kernel<T, NTv, CIVv>), dim3(q.gx), dim3(256), 0, s, g, (const T*)x, (const T*)dy,  \
                       part, bias_part, q.QI, q.QO, q.hch, q.CiP, q.CoP)
        MRI3D_DISPATCH_DTYPE(g.dtype, T, {
            if (g.ci == 1) MRI3D_WGQ(8, 1);
            else if (taps <= 4) MRI3D_WGQ(4, 4);
            else if (taps <= 6) MRI3D_WGQ(6, 4);
            else MRI3D_WGQ(8, 4);
        });
#undef MRI3D_WGQ
        const int total = g.co * g.ci * taps;
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(cdiv(total + g.co, 8)), dim3(256), 0, s, part, bias_part, dw, dbias, q.gx, taps,
                           g.ci, g.co, q.CiP, q.CoP);
        return check_launch("conv3d_wgrad(quads)");
    }
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
        // the kernel's dynamic-LDS limit is raised once (to the CU's 160 KB), not per launch
        static const hipError_t attr_ = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_generic_kernel<T>),
                                                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)attr_;
        hipLaunchKernelGGL(conv_wgrad_generic_kernel<T>, dim3(p.gx, p.taps, p.gz), dim3(256), p.smem, s, g, (const T*)x,
                           (const T*)dy, part, bias_part, p.Ci4, p.Co4, p.nitems, p.vsplit);
    });
    int total = g.co * g.ci * p.taps;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(cdiv(total + g.co, 8)), dim3(256), 0, s, part, bias_part, dw,
                       dbias, p.gx, p.taps, g.ci, g.co, p.Ci4 * 4, p.Co4 * 4);
    return check_launch("conv3d_wgrad(generic)");
}

}  // namespace mri3d
