// norm.hip — BatchNorm3d / InstanceNorm3d statistics, fused normalise+affine+activation forward, and the fused
// backward (activation' -> dgamma/dbeta/dalpha reductions -> dx), NDHWC with voxel pitch.
// Reference semantics: torch.nn.BatchNorm3d (eps 1e-5, momentum .1, biased var for normalisation, unbiased for the
// running estimate) + nn.PReLU / LeakyReLU / ReLU as used by unet.UNet blocks, AE_model.py:30-36, cnn_model.py,
// modified_3dunet.py:20-94 (InstanceNorm3d, affine=False).
//
// Roofline: pure streaming (HBM-bound).  Algorithmic bytes: stats = 1 read of x; fwd apply = 1 read + 1 write;
// bwd = 2 reads (x,dy) for the reductions + 2 reads + 1 write for dx.
//
// Thread mapping: a 256-thread block is a (VT voxels) x (CL channel-lanes) grid, each lane owning VEC consecutive
// channels, so a thread's channels never change across the grid-stride loop: per-channel partial sums live in
// registers, are combined across the VT voxel rows through LDS once per block, and one partial per block goes to
// the workspace.  A finalize kernel sums the partials in a fixed order in double precision (deterministic).
#include "common.h"

namespace mri3d {

struct NormPlan {
    int vec;     // channels per lane (8: bf16 only, 4 or 1)
    int CL;      // channel lanes per voxel row handled by one block (<= 256)
    int VT;      // voxel rows per block iteration
    int cy;      // grid.y = channel chunks
    int nblk;    // grid.x = blocks per group
    int groups;  // grid.z
    int64_t gvox;  // voxels per group
};

constexpr int kNormMaxBlocks = 1024;  // 4 blocks/CU of streaming work; keeps the finalize pass short
constexpr int kFinQL = 64;            // partial-sum lanes per channel in the finalize kernels: with 16 lanes (64 dependent
                                      // loads + adds each over 1024 partials) the two finalize passes took 20 us per launch

static NormPlan norm_plan(const Mri3dNormGeom& g, bool al, bool al16 = false) {
    NormPlan p;
    p.vec = (g.c % 4 == 0 && g.x_ld % 4 == 0 && g.y_ld % 4 == 0 && al) ? 4 : 1;
    // bf16: 8 channels per lane make the accesses 16 bytes wide (with 4 they are 8-byte loads and the streaming kernels sat at
    // 2.9 TB/s against 5.4 TB/s for the same kernels in fp32)
    if (p.vec == 4 && g.dtype == MRI3D_BF16 && g.c % 8 == 0 && g.x_ld % 8 == 0 && g.y_ld % 8 == 0 && al16) p.vec = 8;
    int lanes = g.c / p.vec;
    p.CL = lanes < 256 ? lanes : 256;
    p.cy = cdiv(lanes, p.CL);
    p.VT = 256 / p.CL;
    p.groups = g.instance ? g.n : 1;
    p.gvox = g.instance ? g.vox : (int64_t)g.n * g.vox;
    int64_t want = cdiv64(p.gvox, (int64_t)p.VT * 8);  // >= 8 voxel rows per thread
    int cap = kNormMaxBlocks / (p.groups * p.cy);
    if (cap < 1) cap = 1;
    p.nblk = (int)(want < cap ? want : cap);
    if (p.nblk < 1) p.nblk = 1;
    return p;
}

size_t norm_workspace_floats(const Mri3dNormGeom& g) {
    // plan with worst-case (vec=1) block count is not needed: nblk <= kMaxStreamBlocks/groups always.
    int groups = g.instance ? g.n : 1;
    size_t part = (size_t)(kNormMaxBlocks + groups) * g.c * 3 * 2;  // per-block partials in double (groups*nblk <= kNormMaxBlocks, or nblk=1)
    part += (size_t)groups * g.c * 3;                           // bwd per-(group,channel) sums
    return part;
}

template <int VEC>
struct Ld {
    template <typename T>
    static __device__ __forceinline__ void load(const T* p, float (&v)[VEC]) {
        if constexpr (VEC == 8) {   // bf16 only: one 16-byte load
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
    static __device__ __forceinline__ void store(T* p, const float (&v)[VEC]) {
        if constexpr (VEC == 8) {
            bf16x8_t o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (bf16_t)v[j];
            *reinterpret_cast<bf16x8_t*>(p) = o;
        } else if constexpr (VEC == 4) {
            stf4(p, make_float4(v[0], v[1], v[2], v[3]));
        } else {
            stf(p, v[0]);
        }
    }
};

// ------------------------------------------------------------------ statistics
// partial layout: part[((group*cy... flattened as [group][blk][c][2]
template <typename T, int VEC>
__global__ void __launch_bounds__(256)
norm_stats_kernel(const T* __restrict__ x, double* __restrict__ part, int C, int ld, int64_t gvox, int CL, int VT) {
    __shared__ double red[256 * 2 * VEC];
    const int tid = threadIdx.x;
    const int cl = tid % CL, vt = tid / CL;
    const int c0 = (blockIdx.y * CL + cl) * VEC;
    const bool active = vt < VT && c0 < C;
    const int group = blockIdx.z;
    const T* xg = x + (int64_t)group * gvox * ld;
    // double accumulators: torch's CPU batch-norm (the oracle's arithmetic) accumulates float sums in double
    double s[VEC], ss[VEC];
    float k[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) { s[j] = 0.0; ss[j] = 0.0; k[j] = 0.f; }
    if (active) {
        Ld<VEC>::load(xg + c0, k);  // shift = first voxel of the group: removes E[x^2]-E[x]^2 cancellation
        // four voxel rows per trip: the loads are independent, so four 16-byte requests per lane are in flight (one at a
        // time left this read-only pass at 4.2 TB/s); the sums still run in voxel order
        const int64_t step = (int64_t)gridDim.x * VT;
        int64_t v = (int64_t)blockIdx.x * VT + vt;
        for (; v + 3 * step < gvox; v += 4 * step) {
            float xv[4][VEC];
#pragma unroll
            for (int u = 0; u < 4; ++u) Ld<VEC>::load(xg + (v + u * step) * ld + c0, xv[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    const double d = (double)(xv[u][j] - k[j]);
                    s[j] += d;
                    ss[j] = fma(d, d, ss[j]);
                }
        }
        for (; v < gvox; v += step) {
            float xv[VEC];
            Ld<VEC>::load(xg + v * ld + c0, xv);
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                const double d = (double)(xv[j] - k[j]);
                s[j] += d;
                ss[j] = fma(d, d, ss[j]);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) { red[(tid * VEC + j) * 2] = s[j]; red[(tid * VEC + j) * 2 + 1] = ss[j]; }
    __syncthreads();
    if (vt == 0 && c0 < C) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            double a = 0.0, b = 0.0;
            for (int q = 0; q < VT; ++q) {
                a += red[((q * CL + cl) * VEC + j) * 2];
                b += red[((q * CL + cl) * VEC + j) * 2 + 1];
            }
            double* o = part + (((size_t)group * gridDim.x + blockIdx.x) * C + c0 + j) * 2;
            o[0] = a;
            o[1] = b;
        }
    }
}

template <typename T>
__global__ void __launch_bounds__(256)
norm_stats_finalize_kernel(const T* __restrict__ x, const double* __restrict__ part,
                           float* __restrict__ mean, float* __restrict__ invstd,
                           float* __restrict__ running_mean, float* __restrict__ running_var,
                           float momentum, float eps, int C, int ld, int64_t gvox, int nblk, int groups,
                           const float* __restrict__ shift = nullptr, int use_shift = 0) {
    // 256 threads = 4 (group,channel) slots x kFinQL partial lanes
    __shared__ double ra[256], rb[256];
    const int slot = threadIdx.x / kFinQL, ql = threadIdx.x % kFinQL;
    const int i = blockIdx.x * (256 / kFinQL) + slot;
    const bool ok = i < groups * C;
    const int group = ok ? i / C : 0, c = ok ? i - group * C : 0;
    double a = 0.0, b = 0.0;
    if (ok) {
        const double* p = part + ((size_t)group * nblk * C + c) * 2;
        double a1 = 0.0, b1 = 0.0;   // two independent chains: more partial loads in flight
        int q = ql;
        for (; q + kFinQL < nblk; q += 2 * kFinQL) {
            a += p[(size_t)q * C * 2];
            b += p[(size_t)q * C * 2 + 1];
            a1 += p[(size_t)(q + kFinQL) * C * 2];
            b1 += p[(size_t)(q + kFinQL) * C * 2 + 1];
        }
        if (q < nblk) {
            a += p[(size_t)q * C * 2];
            b += p[(size_t)q * C * 2 + 1];
        }
        a += a1;
        b += b1;
    }
    ra[threadIdx.x] = a;
    rb[threadIdx.x] = b;
    __syncthreads();
    if (!ok || ql != 0) return;
    a = 0.0;
    b = 0.0;
    for (int q = 0; q < kFinQL; ++q) {
        a += ra[slot * kFinQL + q];
        b += rb[slot * kFinQL + q];
    }
    // the partial sums are of (x - k): k = the first voxel's value (norm_stats_kernel), or the caller's per-channel shift (the
    // conv epilogue's sums are of the result without its bias: k = bias, or 0 without one)
    double k = use_shift ? (shift != nullptr ? (double)shift[c] : 0.0) : (double)x[(int64_t)group * gvox * ld + c];
    double cnt = (double)gvox;
    double dm = a / cnt;
    double var = b / cnt - dm * dm;
    if (var < 0.0) var = 0.0;
    double m = k + dm;
    mean[i] = (float)m;
    invstd[i] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean != nullptr && groups == 1) {
        double unb = cnt > 1.0 ? var * cnt / (cnt - 1.0) : var;
        running_mean[c] = (float)((1.0 - momentum) * (double)running_mean[c] + momentum * m);
        running_var[c] = (float)((1.0 - momentum) * (double)running_var[c] + momentum * unb);
    }
}

// GroupNorm statistics: one thread per (n, group) pools the per-channel shifted moments of its group_c channels.
//   per channel: s = sum(x - k), ss = sum((x - k)^2) with k = first voxel  =>  sum x = s + cnt*k, sum x^2 = ss + 2k*s + cnt*k^2
template <typename T>
__global__ void norm_stats_group_finalize_kernel(const T* __restrict__ x, const double* __restrict__ part,
                                                 float* __restrict__ mean, float* __restrict__ invstd, float eps, int C,
                                                 int ld, int64_t gvox, int nblk, int groups, int group_c) {
    const int G = C / group_c;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= groups * G) return;
    const int n = i / G, g = i - n * G;
    double sx = 0.0, sxx = 0.0;
    const double cnt = (double)gvox;
    for (int c = g * group_c; c < (g + 1) * group_c; ++c) {
        double a = 0.0, b = 0.0;
        const double* p = part + ((size_t)n * nblk * C + c) * 2;
        for (int q = 0; q < nblk; ++q) {
            a += p[(size_t)q * C * 2];
            b += p[(size_t)q * C * 2 + 1];
        }
        const double k = (double)x[(int64_t)n * gvox * ld + c];
        sx += a + cnt * k;
        sxx += b + 2.0 * k * a + cnt * k * k;
    }
    const double m = (double)group_c * cnt;
    const double mu = sx / m;
    double var = sxx / m - mu * mu;
    if (var < 0.0) var = 0.0;
    const float fm = (float)mu, fi = (float)(1.0 / sqrt(var + (double)eps));
    for (int c = g * group_c; c < (g + 1) * group_c; ++c) {
        mean[n * C + c] = fm;
        invstd[n * C + c] = fi;
    }
}

// GroupNorm backward: replace the per-(n,c) sums by the gamma-weighted sums of the channel's group (after dgamma/dbeta
// have been taken from the per-channel sums).
__global__ void norm_act_bwd_group_combine_kernel(float* __restrict__ sums, const float* __restrict__ gamma, int C,
                                                  int groups, int group_c) {
    const int G = C / group_c;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= groups * G) return;
    const int n = i / G, g = i - n * G;
    double g0 = 0.0, g1 = 0.0;
    for (int c = g * group_c; c < (g + 1) * group_c; ++c) {
        const double gm = gamma ? (double)gamma[c] : 1.0;
        g0 += gm * (double)sums[((size_t)n * C + c) * 3];
        g1 += gm * (double)sums[((size_t)n * C + c) * 3 + 1];
    }
    for (int c = g * group_c; c < (g + 1) * group_c; ++c) {
        sums[((size_t)n * C + c) * 3] = (float)g0;
        sums[((size_t)n * C + c) * 3 + 1] = (float)g1;
    }
}

// ------------------------------------------------------------------ forward apply
template <typename T, int VEC>
__global__ void __launch_bounds__(256)
norm_act_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, const float* __restrict__ mean,
                    const float* __restrict__ invstd, const float* __restrict__ gamma, const float* __restrict__ beta,
                    const float* __restrict__ alpha, int alpha_n, int act, float slope, int C, int x_ld, int y_ld,
                    int64_t gvox, int CL, int VT) {
    const int tid = threadIdx.x;
    const int cl = tid % CL, vt = tid / CL;
    const int c0 = (blockIdx.y * CL + cl) * VEC;
    if (vt >= VT || c0 >= C) return;
    const int group = blockIdx.z;
    float sc[VEC], sh[VEC], al[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        int c = c0 + j;
        float gm = gamma ? gamma[c] : 1.f;
        float bt = beta ? beta[c] : 0.f;
        float mu = mean ? mean[group * C + c] : 0.f;
        float is = invstd ? invstd[group * C + c] : 1.f;
        sc[j] = gm * is;
        sh[j] = bt - mu * sc[j];
        al[j] = (act == MRI3D_ACT_PRELU) ? alpha[alpha_n == 1 ? 0 : c] : slope;
    }
    const T* xg = x + (int64_t)group * gvox * x_ld;
    T* yg = y + (int64_t)group * gvox * y_ld;
    for (int64_t v = (int64_t)blockIdx.x * VT + vt; v < gvox; v += (int64_t)gridDim.x * VT) {
        float xv[VEC], yv[VEC];
        Ld<VEC>::load(xg + v * x_ld + c0, xv);
#pragma unroll
        for (int j = 0; j < VEC; ++j) yv[j] = apply_act(fmaf(xv[j], sc[j], sh[j]), act, al[j]);
        Ld<VEC>::store(yg + v * y_ld + c0, yv);
    }
}

// ------------------------------------------------------------------ backward: reductions
// part[group][blk][c][3] = (sum du, sum du*xhat, sum dy*u*[u<=0])
template <typename T, int VEC>
__global__ void __launch_bounds__(256)
norm_act_bwd_reduce_kernel(const T* __restrict__ x, const T* __restrict__ dy, double* __restrict__ part,
                           const float* __restrict__ mean, const float* __restrict__ invstd,
                           const float* __restrict__ gamma, const float* __restrict__ beta,
                           const float* __restrict__ alpha, int alpha_n, int act, float slope, int C, int x_ld,
                           int y_ld, int64_t gvox, int CL, int VT) {
    __shared__ double red[256 * 3 * VEC];
    const int tid = threadIdx.x;
    const int cl = tid % CL, vt = tid / CL;
    const int c0 = (blockIdx.y * CL + cl) * VEC;
    const bool active = vt < VT && c0 < C;
    const int group = blockIdx.z;
    double s0[VEC], s1[VEC], s2[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) { s0[j] = 0.0; s1[j] = 0.0; s2[j] = 0.0; }
    if (active) {
        float mu[VEC], is[VEC], gm[VEC], bt[VEC], al[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            int c = c0 + j;
            gm[j] = gamma ? gamma[c] : 1.f;
            bt[j] = beta ? beta[c] : 0.f;
            mu[j] = mean ? mean[group * C + c] : 0.f;
            is[j] = invstd ? invstd[group * C + c] : 1.f;
            al[j] = (act == MRI3D_ACT_PRELU) ? alpha[alpha_n == 1 ? 0 : c]
                                              : (act == MRI3D_ACT_LEAKY ? slope : (act == MRI3D_ACT_RELU ? 0.f : 1.f));
        }
        const T* xg = x + (int64_t)group * gvox * x_ld;
        const T* dg = dy + (int64_t)group * gvox * y_ld;
        auto accumulate = [&](const float (&xv)[VEC], const float (&gv)[VEC]) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                float xh = (xv[j] - mu[j]) * is[j];
                float u = fmaf(gm[j], xh, bt[j]);
                bool pos = u > 0.f;
                float du = pos ? gv[j] : gv[j] * al[j];
                s0[j] += (double)du;
                s1[j] = fma((double)du, (double)xh, s1[j]);
                s2[j] += pos ? 0.0 : (double)gv[j] * (double)u;
            }
        };
        // two voxel rows per trip: four independent 16-byte loads in flight per lane, sums still in voxel order
        const int64_t step = (int64_t)gridDim.x * VT;
        int64_t v = (int64_t)blockIdx.x * VT + vt;
        for (; v + step < gvox; v += 2 * step) {
            float xa[VEC], ga[VEC], xb[VEC], gb[VEC];
            Ld<VEC>::load(xg + v * x_ld + c0, xa);
            Ld<VEC>::load(dg + v * y_ld + c0, ga);
            Ld<VEC>::load(xg + (v + step) * x_ld + c0, xb);
            Ld<VEC>::load(dg + (v + step) * y_ld + c0, gb);
            accumulate(xa, ga);
            accumulate(xb, gb);
        }
        for (; v < gvox; v += step) {
            float xv[VEC], gv[VEC];
            Ld<VEC>::load(xg + v * x_ld + c0, xv);
            Ld<VEC>::load(dg + v * y_ld + c0, gv);
            accumulate(xv, gv);
        }
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        red[(tid * VEC + j) * 3] = s0[j];
        red[(tid * VEC + j) * 3 + 1] = s1[j];
        red[(tid * VEC + j) * 3 + 2] = s2[j];
    }
    __syncthreads();
    if (vt == 0 && c0 < C) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            double a = 0.0, b = 0.0, d = 0.0;
            for (int q = 0; q < VT; ++q) {
                a += red[((q * CL + cl) * VEC + j) * 3];
                b += red[((q * CL + cl) * VEC + j) * 3 + 1];
                d += red[((q * CL + cl) * VEC + j) * 3 + 2];
            }
            double* o = part + (((size_t)group * gridDim.x + blockIdx.x) * C + c0 + j) * 3;
            o[0] = a;
            o[1] = b;
            o[2] = d;
        }
    }
}

// Stage A: sums[group][c][3] = fixed-order double sums of the per-block partials.
__global__ void __launch_bounds__(256)
norm_act_bwd_sums_kernel(const double* __restrict__ part, float* __restrict__ sums, int C, int nblk, int groups) {
    __shared__ double r0[256], r1[256], r2[256];
    const int slot = threadIdx.x / kFinQL, ql = threadIdx.x % kFinQL;
    const int i = blockIdx.x * (256 / kFinQL) + slot;
    const bool ok = i < groups * C;
    const int group = ok ? i / C : 0, c = ok ? i - group * C : 0;
    double a = 0.0, b = 0.0, d = 0.0;
    if (ok) {
        const double* p = part + ((size_t)group * nblk * C + c) * 3;
        double a1 = 0.0, b1 = 0.0, d1 = 0.0;
        int q = ql;
        for (; q + kFinQL < nblk; q += 2 * kFinQL) {
            a += p[(size_t)q * C * 3];
            b += p[(size_t)q * C * 3 + 1];
            d += p[(size_t)q * C * 3 + 2];
            a1 += p[(size_t)(q + kFinQL) * C * 3];
            b1 += p[(size_t)(q + kFinQL) * C * 3 + 1];
            d1 += p[(size_t)(q + kFinQL) * C * 3 + 2];
        }
        if (q < nblk) {
            a += p[(size_t)q * C * 3];
            b += p[(size_t)q * C * 3 + 1];
            d += p[(size_t)q * C * 3 + 2];
        }
        a += a1;
        b += b1;
        d += d1;
    }
    r0[threadIdx.x] = a;
    r1[threadIdx.x] = b;
    r2[threadIdx.x] = d;
    __syncthreads();
    if (!ok || ql != 0) return;
    a = b = d = 0.0;
    for (int q = 0; q < kFinQL; ++q) {
        a += r0[slot * kFinQL + q];
        b += r1[slot * kFinQL + q];
        d += r2[slot * kFinQL + q];
    }
    sums[(size_t)i * 3] = (float)a;
    sums[(size_t)i * 3 + 1] = (float)b;
    sums[(size_t)i * 3 + 2] = (float)d;
}

// Stage B (one block): dbeta/dgamma = sum over groups; dalpha per channel or grand total.  Run by the first workgroup of the apply
// kernel (batch / instance norm: one launch less per backward, 5 us each, ten per step of the U-Net) or, for GroupNorm — whose
// combine step rewrites `sums` before the apply kernel — as the kernel below.
__device__ __forceinline__ void norm_act_bwd_params_block(const float* __restrict__ sums, float* __restrict__ dgamma,
                                                          float* __restrict__ dbeta, float* __restrict__ dalpha, int alpha_n, int C,
                                                          int groups) {
    __shared__ double dal[256];
    double my_dal = 0.0;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        double tg = 0.0, tb = 0.0, ta = 0.0;
        for (int g = 0; g < groups; ++g) {
            tb += (double)sums[((size_t)g * C + c) * 3];
            tg += (double)sums[((size_t)g * C + c) * 3 + 1];
            ta += (double)sums[((size_t)g * C + c) * 3 + 2];
        }
        if (dgamma) dgamma[c] = (float)tg;
        if (dbeta) dbeta[c] = (float)tb;
        if (dalpha && alpha_n > 1) dalpha[c] = (float)ta;
        my_dal += ta;
    }
    if (dalpha && alpha_n == 1) {
        dal[threadIdx.x] = my_dal;
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = 0.0;
            for (int i = 0; i < (int)blockDim.x; ++i) t += dal[i];
            dalpha[0] = (float)t;
        }
    }
}

__global__ void norm_act_bwd_params_kernel(const float* __restrict__ sums, float* __restrict__ dgamma,
                                           float* __restrict__ dbeta, float* __restrict__ dalpha, int alpha_n, int C,
                                           int groups) {
    norm_act_bwd_params_block(sums, dgamma, dbeta, dalpha, alpha_n, C, groups);
}

// ------------------------------------------------------------------ backward: dx
template <typename T, int VEC>
__global__ void __launch_bounds__(256)
norm_act_bwd_apply_kernel(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx,
                          const float* __restrict__ sums, const float* __restrict__ mean,
                          const float* __restrict__ invstd, const float* __restrict__ gamma,
                          const float* __restrict__ beta, const float* __restrict__ alpha, int alpha_n, int act,
                          float slope, int training, int C, int x_ld, int y_ld, int64_t gvox, int CL, int VT, int group_c,
                          float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ dalpha) {
    // the parameter gradients (stage B) ride on the first workgroup; every pointer null = already done / not wanted (uniform)
    if ((dgamma != nullptr || dbeta != nullptr || dalpha != nullptr) && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0)
        norm_act_bwd_params_block(sums, dgamma, dbeta, dalpha, alpha_n, C, (int)gridDim.z);
    const int tid = threadIdx.x;
    const int cl = tid % CL, vt = tid / CL;
    const int c0 = (blockIdx.y * CL + cl) * VEC;
    if (vt >= VT || c0 >= C) return;
    const int group = blockIdx.z;
    float mu[VEC], is[VEC], gm[VEC], bt[VEC], al[VEC], k0[VEC], k1[VEC], k2[VEC];
    const float invM = 1.f / ((float)gvox * (float)(group_c > 0 ? group_c : 1));
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        int c = c0 + j;
        gm[j] = gamma ? gamma[c] : 1.f;
        bt[j] = beta ? beta[c] : 0.f;
        mu[j] = mean ? mean[group * C + c] : 0.f;
        is[j] = invstd ? invstd[group * C + c] : 1.f;
        al[j] = (act == MRI3D_ACT_PRELU) ? alpha[alpha_n == 1 ? 0 : c]
                                          : (act == MRI3D_ACT_LEAKY ? slope : (act == MRI3D_ACT_RELU ? 0.f : 1.f));
        // dx = k0*du - k1 - xhat*k2
        k0[j] = gm[j] * is[j];
        if (training) {
            // batch / instance norm: sums are per channel and scale with gamma*invstd; group norm: sums already hold the
            // gamma-weighted totals of the channel's group and scale with invstd only
            const float kk = group_c > 0 ? is[j] : k0[j];
            k1[j] = kk * sums[((size_t)group * C + c) * 3] * invM;
            k2[j] = kk * sums[((size_t)group * C + c) * 3 + 1] * invM;
        } else {
            k1[j] = 0.f;
            k2[j] = 0.f;
        }
    }
    const T* xg = x + (int64_t)group * gvox * x_ld;
    const T* dg = dy + (int64_t)group * gvox * y_ld;
    T* og = dx + (int64_t)group * gvox * x_ld;
    auto apply = [&](const float (&xv)[VEC], const float (&gv)[VEC], float (&ov)[VEC]) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            float xh = (xv[j] - mu[j]) * is[j];
            float u = fmaf(gm[j], xh, bt[j]);
            float du = (u > 0.f) ? gv[j] : gv[j] * al[j];
            ov[j] = fmaf(k0[j], du, -k1[j]) - xh * k2[j];
        }
    };
    // two voxel rows per trip: four independent 16-byte loads in flight per lane
    const int64_t step = (int64_t)gridDim.x * VT;
    int64_t v = (int64_t)blockIdx.x * VT + vt;
    for (; v + step < gvox; v += 2 * step) {
        float xa[VEC], ga[VEC], xb[VEC], gb[VEC], ov[VEC];
        Ld<VEC>::load(xg + v * x_ld + c0, xa);
        Ld<VEC>::load(dg + v * y_ld + c0, ga);
        Ld<VEC>::load(xg + (v + step) * x_ld + c0, xb);
        Ld<VEC>::load(dg + (v + step) * y_ld + c0, gb);
        apply(xa, ga, ov);
        Ld<VEC>::store(og + v * x_ld + c0, ov);
        apply(xb, gb, ov);
        Ld<VEC>::store(og + (v + step) * x_ld + c0, ov);
    }
    for (; v < gvox; v += step) {
        float xv[VEC], gv[VEC], ov[VEC];
        Ld<VEC>::load(xg + v * x_ld + c0, xv);
        Ld<VEC>::load(dg + v * y_ld + c0, gv);
        apply(xv, gv, ov);
        Ld<VEC>::store(og + v * x_ld + c0, ov);
    }
}

}  // namespace mri3d

using namespace mri3d;

extern "C" size_t mri3d_norm_workspace_bytes(const Mri3dNormGeom* g) {
    if (!g) return 0;
    return norm_workspace_floats(*g) * sizeof(float);
}

static int norm_check(const Mri3dNormGeom* g, const char* who) {
    MRI3D_REQUIRE(g != nullptr, MRI3D_EINVAL, "%s: null geometry", who);
    MRI3D_REQUIRE(g->dtype == MRI3D_F32 || g->dtype == MRI3D_BF16, MRI3D_ENOTSUP, "%s: unknown dtype %d", who, g->dtype);
    MRI3D_REQUIRE(g->n > 0 && g->vox > 0 && g->c > 0 && g->x_ld >= g->c && g->y_ld >= g->c, MRI3D_EINVAL,
                  "%s: bad geometry n=%d vox=%lld c=%d x_ld=%d y_ld=%d", who, g->n, (long long)g->vox, g->c, g->x_ld,
                  g->y_ld);
    MRI3D_REQUIRE(g->act != MRI3D_ACT_PRELU || g->alpha_n == 1 || g->alpha_n == g->c, MRI3D_EINVAL,
                  "%s: PReLU alpha_n=%d must be 1 or C=%d", who, g->alpha_n, g->c);
    MRI3D_REQUIRE(g->group_c == 0 || (g->group_c > 0 && g->instance && g->c % g->group_c == 0), MRI3D_EINVAL,
                  "%s: group_c=%d needs instance mode and must divide C=%d", who, g->group_c, g->c);
    return MRI3D_OK;
}

extern "C" int mri3d_norm_stats(const Mri3dNormGeom* g, const void* x, float* mean, float* invstd, float* running_mean,
                                float* running_var, float momentum, void* workspace, size_t ws_bytes,
                                mri3d_stream_t stream) {
    int rc = norm_check(g, "norm_stats");
    if (rc) return rc;
    MRI3D_REQUIRE(x && mean && invstd, MRI3D_EINVAL, "norm_stats: null pointer");
    MRI3D_REQUIRE(workspace && ws_bytes >= mri3d_norm_workspace_bytes(g), MRI3D_EWORKSPACE,
                  "norm_stats: workspace %zu < %zu", ws_bytes, mri3d_norm_workspace_bytes(g));
    hipStream_t s = static_cast<hipStream_t>(stream);
    Mri3dNormGeom gg = *g;
    gg.y_ld = gg.x_ld;
    NormPlan p = norm_plan(gg, aligned_vec4(g->dtype, x));   // 8 channels per lane measured slower here (float64 accumulators)
    double* part = static_cast<double*>(workspace);
    dim3 grid(p.nblk, p.cy, p.groups);
    const int tot = p.groups * g->c;
    MRI3D_DISPATCH_DTYPE(g->dtype, T, {
        const T* xf = static_cast<const T*>(x);
        if (p.vec == 4)
            hipLaunchKernelGGL((norm_stats_kernel<T, 4>), grid, dim3(256), 0, s, xf, part, g->c, g->x_ld, p.gvox, p.CL, p.VT);
        else if (p.vec == 1)
            hipLaunchKernelGGL((norm_stats_kernel<T, 1>), grid, dim3(256), 0, s, xf, part, g->c, g->x_ld, p.gvox, p.CL, p.VT);
        hipLaunchKernelGGL(norm_stats_finalize_kernel<T>, dim3(cdiv(tot, 256 / kFinQL)), dim3(256), 0, s, xf, part, mean,
                           invstd, running_mean, running_var, momentum, g->eps, g->c, g->x_ld, p.gvox, p.nblk, p.groups);
        if (g->group_c > 0) {
            const int ng = p.groups * (g->c / g->group_c);
            hipLaunchKernelGGL(norm_stats_group_finalize_kernel<T>, dim3(cdiv(ng, 64)), dim3(64), 0, s, xf, part, mean,
                               invstd, g->eps, g->c, g->x_ld, p.gvox, p.nblk, p.groups, g->group_c);
        }
    });
    return check_launch("norm_stats");
}

extern "C" int mri3d_norm_stats_from_partials(const Mri3dNormGeom* g, const double* partials, int32_t nblk, const float* shift,
                                              float* mean, float* invstd, float* running_mean, float* running_var,
                                              float momentum, mri3d_stream_t stream) {
    int rc = norm_check(g, "norm_stats_from_partials");
    if (rc) return rc;
    MRI3D_REQUIRE(partials && mean && invstd && nblk > 0, MRI3D_EINVAL, "norm_stats_from_partials: null pointer / no partials");
    MRI3D_REQUIRE(!g->instance && g->group_c == 0, MRI3D_ENOTSUP,
                  "norm_stats_from_partials: batch statistics only (one group over N x voxels)");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int64_t gvox = (int64_t)g->n * g->vox;
    hipLaunchKernelGGL(norm_stats_finalize_kernel<float>, dim3(cdiv(g->c, 256 / kFinQL)), dim3(256), 0, s,
                       static_cast<const float*>(nullptr), partials, mean, invstd, running_mean, running_var, momentum, g->eps,
                       g->c, g->x_ld, gvox, nblk, 1, shift, 1);
    return check_launch("norm_stats_from_partials");
}

extern "C" int mri3d_norm_act_fwd(const Mri3dNormGeom* g, const void* x, const float* mean, const float* invstd,
                                  const float* gamma, const float* beta, const float* alpha, void* y,
                                  mri3d_stream_t stream) {
    int rc = norm_check(g, "norm_act_fwd");
    if (rc) return rc;
    MRI3D_REQUIRE(x && y, MRI3D_EINVAL, "norm_act_fwd: null pointer");
    MRI3D_REQUIRE((mean == nullptr) == (invstd == nullptr), MRI3D_EINVAL, "norm_act_fwd: mean/invstd must both be set");
    MRI3D_REQUIRE(g->act != MRI3D_ACT_PRELU || alpha, MRI3D_EINVAL, "norm_act_fwd: PReLU needs alpha");
    hipStream_t s = static_cast<hipStream_t>(stream);
    NormPlan p = norm_plan(*g, aligned_vec4(g->dtype, x, y), aligned16(x, y));
    dim3 grid(p.nblk, p.cy, p.groups);
    MRI3D_DISPATCH_DTYPE(g->dtype, T, {
        const T* xf = static_cast<const T*>(x);
        T* yf = static_cast<T*>(y);
        if constexpr (sizeof(T) == 2) {
            if (p.vec == 8) hipLaunchKernelGGL((norm_act_fwd_kernel<T, 8>), grid, dim3(256), 0, s, xf, yf, mean, invstd, gamma, beta, alpha,
                               g->alpha_n, g->act, g->slope, g->c, g->x_ld, g->y_ld, p.gvox, p.CL, p.VT);
        }
        if (p.vec == 4)
            hipLaunchKernelGGL((norm_act_fwd_kernel<T, 4>), grid, dim3(256), 0, s, xf, yf, mean, invstd, gamma, beta, alpha,
                               g->alpha_n, g->act, g->slope, g->c, g->x_ld, g->y_ld, p.gvox, p.CL, p.VT);
        else if (p.vec == 1)
            hipLaunchKernelGGL((norm_act_fwd_kernel<T, 1>), grid, dim3(256), 0, s, xf, yf, mean, invstd, gamma, beta, alpha,
                               g->alpha_n, g->act, g->slope, g->c, g->x_ld, g->y_ld, p.gvox, p.CL, p.VT);
    });
    return check_launch("norm_act_fwd");
}

extern "C" int mri3d_norm_act_bwd(const Mri3dNormGeom* g, int training, const void* x, const void* dy,
                                  const float* mean, const float* invstd, const float* gamma, const float* beta,
                                  const float* alpha, void* dx, float* dgamma, float* dbeta, float* dalpha,
                                  void* workspace, size_t ws_bytes, mri3d_stream_t stream) {
    int rc = norm_check(g, "norm_act_bwd");
    if (rc) return rc;
    MRI3D_REQUIRE(x && dy && dx, MRI3D_EINVAL, "norm_act_bwd: null pointer");
    MRI3D_REQUIRE((mean == nullptr) == (invstd == nullptr), MRI3D_EINVAL, "norm_act_bwd: mean/invstd must both be set");
    MRI3D_REQUIRE(!(training && mean == nullptr), MRI3D_EINVAL, "norm_act_bwd: training mode needs statistics");
    MRI3D_REQUIRE(g->act != MRI3D_ACT_PRELU || alpha, MRI3D_EINVAL, "norm_act_bwd: PReLU needs alpha");
    MRI3D_REQUIRE(workspace && ws_bytes >= mri3d_norm_workspace_bytes(g), MRI3D_EWORKSPACE,
                  "norm_act_bwd: workspace %zu < %zu", ws_bytes, mri3d_norm_workspace_bytes(g));
    hipStream_t s = static_cast<hipStream_t>(stream);
    // x/dx share pitch x_ld, dy has pitch y_ld
    NormPlan p = norm_plan(*g, aligned_vec4(g->dtype, x, dy, dx));   // 8 channels per lane: reduce 66 -> 101 us, apply 74 -> 83 us (bf16 bench): not used
    dim3 grid(p.nblk, p.cy, p.groups);
    double* part = static_cast<double*>(workspace);
    float* sums = reinterpret_cast<float*>(part + (size_t)p.groups * p.nblk * g->c * 3);
    const bool need_reduce = training || dgamma || dbeta || dalpha;
    bool params_done = false;
    MRI3D_DISPATCH_DTYPE(g->dtype, T, {
        const T* xf = static_cast<const T*>(x);
        const T* df = static_cast<const T*>(dy);
        T* of = static_cast<T*>(dx);
        if (need_reduce) {
            if (p.vec == 4)
                hipLaunchKernelGGL((norm_act_bwd_reduce_kernel<T, 4>), grid, dim3(256), 0, s, xf, df, part, mean, invstd,
                                   gamma, beta, alpha, g->alpha_n, g->act, g->slope, g->c, g->x_ld, g->y_ld, p.gvox, p.CL,
                                   p.VT);
            else if (p.vec == 1)
                hipLaunchKernelGGL((norm_act_bwd_reduce_kernel<T, 1>), grid, dim3(256), 0, s, xf, df, part, mean, invstd,
                                   gamma, beta, alpha, g->alpha_n, g->act, g->slope, g->c, g->x_ld, g->y_ld, p.gvox, p.CL,
                                   p.VT);
            hipLaunchKernelGGL(norm_act_bwd_sums_kernel, dim3(cdiv(p.groups * g->c, 256 / kFinQL)), dim3(256), 0, s, part,
                               sums, g->c, p.nblk, p.groups);
            if ((dgamma || dbeta || dalpha) && g->group_c > 0) {   // GroupNorm: before the combine step rewrites `sums`
                hipLaunchKernelGGL(norm_act_bwd_params_kernel, dim3(1), dim3(256), 0, s, sums, dgamma, dbeta, dalpha,
                                   g->alpha_n, g->c, p.groups);
                params_done = true;
            }
            if (g->group_c > 0 && training) {
                const int ng = p.groups * (g->c / g->group_c);
                hipLaunchKernelGGL(norm_act_bwd_group_combine_kernel, dim3(cdiv(ng, 64)), dim3(64), 0, s, sums, gamma,
                                   g->c, p.groups, g->group_c);
            }
        }
        float* pg = params_done ? nullptr : dgamma;
        float* pb = params_done ? nullptr : dbeta;
        float* pa = params_done ? nullptr : dalpha;
        if (p.vec == 4)
            hipLaunchKernelGGL((norm_act_bwd_apply_kernel<T, 4>), grid, dim3(256), 0, s, xf, df, of, sums, mean, invstd,
                               gamma, beta, alpha, g->alpha_n, g->act, g->slope, training, g->c, g->x_ld, g->y_ld, p.gvox,
                               p.CL, p.VT, g->group_c, pg, pb, pa);
        else if (p.vec == 1)
            hipLaunchKernelGGL((norm_act_bwd_apply_kernel<T, 1>), grid, dim3(256), 0, s, xf, df, of, sums, mean, invstd,
                               gamma, beta, alpha, g->alpha_n, g->act, g->slope, training, g->c, g->x_ld, g->y_ld, p.gvox,
                               p.CL, p.VT, g->group_c, pg, pb, pa);
    });
    return check_launch("norm_act_bwd");
}
