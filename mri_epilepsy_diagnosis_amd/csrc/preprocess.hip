// preprocess.hip — intensity standardisation ahead of the conv path: the histogram standardisation the
// reference applies in its collate function (`normalize`, classification/train_ENC_CLF.ipynb cell 9: np.percentile over the
// whole volume at 13 percentiles, then an 10-segment piecewise-linear map in float64) — in the reference this is
// np.percentile + np.digitize over ~5-7 M voxels per volume on one CPU core per sample.
//
//   mri3d_order_stats_f32     exact order statistics x_(r) for up to 32 ranks of an fp32 array: three radix passes
//                             (11 + 11 + 10 key bits) of integer histograms.  Integer atomics only, so the result is exact
//                             and run-to-run identical; np.percentile's linear interpolation between two neighbouring order
//                             statistics is then 13 float64 operations on the host, bit-identical to numpy's _lerp.
//   mri3d_piecewise_linear_f32  y = float32(slope[b] * double(x) + intercept[b]),  b = #(edges <= x)  (np.digitize, right=False),
//                             with separately rounded multiply and add like numpy's `lin_img * data + aff_img`.
// Both are HBM-bound streams (4 B/voxel read per radix pass; 8 B/voxel for the map).
#include "common.h"
#include <algorithm>

namespace mri3d {

constexpr int kMaxRanks = 32;
constexpr int kB0 = 2048, kB1 = 2048, kB2 = 1024;   // bins of the three radix levels: key bits [31:21], [20:10], [9:0]

struct RankSet { long long rank[kMaxRanks]; int n; };
struct PwlMap { double edge[16]; double slope[16]; double icpt[16]; int nseg; };

// order-preserving key: ascending unsigned order == ascending float order (-0.0 sorts just below +0.0, NaN payloads last)
__device__ __forceinline__ unsigned f32_key(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_f32(unsigned k) {
    const unsigned u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(u);
}

// Histogram increment with wave-level aggregation.  Skull-stripped volumes are ~45 % exact zeros: every lane hitting the
// same counter would serialise (50 ms per volume measured with plain atomics).  Two rounds of "leader's bin": the lanes that
// share the first pending lane's bin are counted with one popcount-sized add; whatever is left falls back to single adds.
// Must be called by all lanes of the wave (`active` masks the tail).
__device__ __forceinline__ void agg_hist_add(unsigned* hist, unsigned idx, bool active) {
    const int lane = threadIdx.x & 63;
    unsigned long long todo = __ballot(active);
#pragma unroll
    for (int round = 0; round < 2; ++round) {
        if (todo == 0ull) break;   // wave-uniform
        const int leader = __ffsll((long long)todo) - 1;
        const unsigned lidx = (unsigned)__shfl((int)idx, leader, 64);
        const bool same = active && idx == lidx;
        const unsigned long long m = __ballot(same);
        if (lane == leader) atomicAdd(&hist[lidx], (unsigned)__popcll(m));
        active = active && !same;
        todo = __ballot(active);
    }
    if (active) atomicAdd(&hist[idx], 1u);
}

// state layout in the workspace (all unsigned 32-bit unless noted):
//   remaining[R] (u64) | hist0[kB0] | hist1[R][kB1] | hist2[R][kB2] | prefix[R]
__global__ void __launch_bounds__(256)
ostat_hist0_kernel(const float* __restrict__ x, long long n, unsigned* __restrict__ hist0) {
    __shared__ unsigned h[kB0];
    for (int i = threadIdx.x; i < kB0; i += blockDim.x) h[i] = 0u;
    __syncthreads();
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long base = (long long)blockIdx.x * blockDim.x; base < n; base += stride) {
        const long long i = base + threadIdx.x;
        const bool active = i < n;
        agg_hist_add(h, active ? f32_key(x[i]) >> 21 : 0u, active);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kB0; i += blockDim.x)
        if (h[i]) atomicAdd(&hist0[i], h[i]);
}

// level 1 / 2: only elements whose leading bits equal some rank's prefix are counted (about R/2048 of the data — or all
// the background voxels when an order statistic IS the background value), straight into global memory
// Block-private counting table for levels 1 / 2: (global bin index -> count) in LDS, open addressing, flushed once per block.
// When an order statistic is the background value, millions of voxels land on ONE global counter; through the table they
// become LDS adds plus one global add per block.
constexpr int kHT = 1024;
__device__ __forceinline__ void ht_add(unsigned* keys, unsigned* cnts, unsigned* hist, unsigned idx, unsigned c) {
    unsigned slot = (idx * 2654435761u) >> 22;   // top 10 bits of a Fibonacci hash
    for (int probe = 0; probe < 8; ++probe) {
        const unsigned prev = atomicCAS(&keys[slot], 0xffffffffu, idx);
        if (prev == 0xffffffffu || prev == idx) {
            atomicAdd(&cnts[slot], c);
            return;
        }
        slot = (slot + 1) & (kHT - 1);
    }
    atomicAdd(&hist[idx], c);   // table crowded around this slot: straight to global memory
}

template <int LEVEL>
__global__ void __launch_bounds__(256)
ostat_hist_kernel(const float* __restrict__ x, long long n, const unsigned* __restrict__ prefix, unsigned* __restrict__ hist,
                  int R) {
    __shared__ unsigned pf[kMaxRanks];
    __shared__ unsigned hkeys[kHT], hcnts[kHT];
    if (threadIdx.x < R) pf[threadIdx.x] = prefix[threadIdx.x];
    for (int i = threadIdx.x; i < kHT; i += blockDim.x) { hkeys[i] = 0xffffffffu; hcnts[i] = 0u; }
    __syncthreads();
    constexpr int SH = LEVEL == 1 ? 21 : 10;
    const int lane = threadIdx.x & 63;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long base = (long long)blockIdx.x * blockDim.x; base < n; base += stride) {
        const long long i = base + threadIdx.x;
        bool active = i < n;
        const unsigned k = active ? f32_key(x[i]) : 0u;
        const unsigned top = k >> SH;
        unsigned idx = 0u;
        bool hit = false;
        for (int r = 0; r < R; ++r) {   // ranks sharing a prefix share the histogram row of the first of them
            if (!hit && pf[r] == top) {
                idx = LEVEL == 1 ? (unsigned)r * kB1 + ((k >> 10) & 0x7ffu) : (unsigned)r * kB2 + (k & 0x3ffu);
                hit = true;
            }
        }
        active = active && hit;
        // wave-level aggregation (two leader rounds), each aggregate goes through the block's table
        unsigned long long todo = __ballot(active);
#pragma unroll
        for (int round = 0; round < 2; ++round) {
            if (todo == 0ull) break;
            const int leader = __ffsll((long long)todo) - 1;
            const unsigned lidx = (unsigned)__shfl((int)idx, leader, 64);
            const bool same = active && idx == lidx;
            const unsigned long long m = __ballot(same);
            if (lane == leader) ht_add(hkeys, hcnts, hist, lidx, (unsigned)__popcll(m));
            active = active && !same;
            todo = __ballot(active);
        }
        if (active) ht_add(hkeys, hcnts, hist, idx, 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kHT; i += blockDim.x)
        if (hkeys[i] != 0xffffffffu && hcnts[i]) atomicAdd(&hist[hkeys[i]], hcnts[i]);
}

// one thread per rank walks its histogram row: bin whose cumulative count passes `remaining`
template <int LEVEL>
__global__ void ostat_select_kernel(RankSet rs, const unsigned* __restrict__ hist, unsigned* __restrict__ prefix,
                                    unsigned long long* __restrict__ remaining, float* __restrict__ out) {
    const int r = threadIdx.x;
    if (r >= rs.n) return;
    constexpr int NB = LEVEL == 0 ? kB0 : (LEVEL == 1 ? kB1 : kB2);
    int row = 0;
    if (LEVEL > 0) {   // the row of the first rank with the same prefix
        row = r;
        for (int q = 0; q < r; ++q)
            if (prefix[q] == prefix[r]) { row = q; break; }
    }
    const unsigned* h = hist + (size_t)row * NB;
    unsigned long long rem = LEVEL == 0 ? (unsigned long long)rs.rank[r] : remaining[r];
    int b = 0;
    for (; b < NB - 1; ++b) {
        const unsigned c = h[b];
        if (rem < c) break;
        rem -= c;
    }
    __syncthreads();   // every lane has read the shared prefix table before anyone rewrites it
    const unsigned np = LEVEL == 0 ? (unsigned)b : ((prefix[r] << (LEVEL == 1 ? 11 : 10)) | (unsigned)b);
    prefix[r] = np;
    remaining[r] = rem;
    if (LEVEL == 2) out[r] = key_f32(np);
}

__global__ void __launch_bounds__(256)
pwl_kernel(const float* __restrict__ x, float* __restrict__ y, long long n, PwlMap m) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const double xd = (double)x[i];
        int b = 0;
#pragma unroll
        for (int e = 0; e < 15; ++e) b += (e < m.nseg - 1 && m.edge[e] <= xd) ? 1 : 0;
        // numpy: lin_img * data (rounded) + aff_img (rounded) — no fused multiply-add
        y[i] = (float)__dadd_rn(__dmul_rn(m.slope[b], xd), m.icpt[b]);
    }
}

// ------------------------------------------------------------------ z-normalisation over a foreground mask
// TorchIO's ZNormalization(masking_method=mean) as the notebooks configure it (segmentation/pretraining_3d_unet.ipynb cell
// 8, `ZNormalization(masking_method=ZNormalization.mean)`): mask = x > mean(x); y = (x - mean(x[mask])) / std(x[mask]) with
// the unbiased (N-1) standard deviation.  Third-party arithmetic (TorchIO, version not pinned by the reference): restated
// from its documentation — "parity unpinned".  Two reduction passes in double with fixed-order block partials.
constexpr int kZBlocks = 512;

// pass 0: part[b] = (sum, 0, count=all);  pass 1: over x > thr: part[b] = (sum (x - shift), sum (x - shift)^2, count)
__global__ void __launch_bounds__(256)
znorm_partial_kernel(const float* __restrict__ x, long long n, int masked, const double* __restrict__ stats,
                     double* __restrict__ part) {
    __shared__ double red[4][3];
    // the threshold is compared in float32 like `tensor > tensor.mean()`; the shift only conditions the sums
    const float thr = masked ? (float)stats[0] : 0.f;
    const double shift = masked ? stats[0] : 0.0;
    double s = 0.0, ss = 0.0, c = 0.0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float v = x[i];
        if (!masked || v > thr) {
            const double d = (double)v - shift;
            s += d;
            ss = fma(d, d, ss);
            c += 1.0;
        }
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    s = wave_sum_d(s);
    ss = wave_sum_d(ss);
    c = wave_sum_d(c);
    if (lane == 0) { red[wave][0] = s; red[wave][1] = ss; red[wave][2] = c; }
    __syncthreads();
    if (threadIdx.x < 3)
        part[(size_t)blockIdx.x * 3 + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// stats[0] = mean of all voxels; stats[1] = masked count; stats[2] = masked mean; stats[3] = masked unbiased std
__global__ void znorm_finalize_kernel(const double* __restrict__ part, int nblk, int masked, double* __restrict__ stats) {
    if (threadIdx.x != 0) return;
    double s = 0.0, ss = 0.0, c = 0.0;
    for (int b = 0; b < nblk; ++b) { s += part[b * 3]; ss += part[b * 3 + 1]; c += part[b * 3 + 2]; }
    if (!masked) {
        stats[0] = (double)(float)(s / c);   // tensor.mean() is a float32 tensor: the mask threshold is its float32 value
        return;
    }
    const double shift = stats[0];
    const double m = c > 0.0 ? s / c : 0.0;
    const double var = c > 1.0 ? (ss - s * m) / (c - 1.0) : 0.0;
    stats[1] = c;
    stats[2] = shift + m;
    stats[3] = sqrt(var > 0.0 ? var : 0.0);
}

// y = (x - mean) / std in float32, subtraction and division rounded separately like `(tensor - mean) / std`
__global__ void __launch_bounds__(256)
znorm_apply_kernel(const float* __restrict__ x, float* __restrict__ y, long long n, const double* __restrict__ stats) {
    const float mean = (float)stats[2], sd = (float)stats[3];
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        y[i] = __fdiv_rn(__fsub_rn(x[i], mean), sd);
}

// ------------------------------------------------------------------ centred crop-or-pad
// TorchIO's CropOrPad(target_shape) (segmentation/pretraining_3d_unet.ipynb cell 9): per axis the difference is split
// floor/ceil between the two ends (crop: ini = floor(diff/2); pad: ini = floor(diff/2) of zeros in front).  One gather per
// output voxel; `outer` leading (batch x channel) volumes.  Third-party semantics — "parity unpinned".
__global__ void __launch_bounds__(256)
crop_or_pad_kernel(const float* __restrict__ x, float* __restrict__ y, int outer, int di, int hi, int wi, int dout, int ho,
                   int wo, int od, int oh, int ow, float fill) {
    // (od, oh, ow) = offset of the output origin in input coordinates (negative = padding in front)
    const long long per = (long long)dout * ho * wo, total = per * outer;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int o = (int)(i / per);
        unsigned r = (unsigned)(i - (long long)o * per);
        const int w = r % wo;
        r /= wo;
        const int h = r % ho, d = r / ho;
        const int id = d + od, ih = h + oh, iw = w + ow;
        const bool in = (unsigned)id < (unsigned)di && (unsigned)ih < (unsigned)hi && (unsigned)iw < (unsigned)wi;
        y[i] = in ? x[(((long long)o * di + id) * hi + ih) * wi + iw] : fill;
    }
}

}  // namespace mri3d

using namespace mri3d;

extern "C" size_t mri3d_znorm_workspace_bytes(void) { return (size_t)kZBlocks * 3 * sizeof(double); }

extern "C" int mri3d_znorm_mean_mask_f32(const float* x, float* y, int64_t n, double* stats, void* workspace,
                                         size_t ws_bytes, mri3d_stream_t stream) {
    MRI3D_REQUIRE(x && y && stats && n > 0, MRI3D_EINVAL, "znorm: bad arguments");
    MRI3D_REQUIRE(workspace && ws_bytes >= mri3d_znorm_workspace_bytes() && (reinterpret_cast<uintptr_t>(workspace) & 7) == 0,
                  MRI3D_EWORKSPACE, "znorm: workspace %zu < %zu", ws_bytes, mri3d_znorm_workspace_bytes());
    hipStream_t s = static_cast<hipStream_t>(stream);
    double* part = static_cast<double*>(workspace);
    int nblk = (int)std::min<int64_t>(kZBlocks, cdiv64(n, 256 * 8));
    if (nblk < 1) nblk = 1;
    for (int masked = 0; masked < 2; ++masked) {
        hipLaunchKernelGGL(znorm_partial_kernel, dim3(nblk), dim3(256), 0, s, x, (long long)n, masked, stats, part);
        hipLaunchKernelGGL(znorm_finalize_kernel, dim3(1), dim3(64), 0, s, part, nblk, masked, stats);
    }
    hipLaunchKernelGGL(znorm_apply_kernel, dim3(stream_grid(n, 256 * 4)), dim3(256), 0, s, x, y, (long long)n, stats);
    return check_launch("znorm");
}

extern "C" int mri3d_crop_or_pad_f32(const float* x, float* y, int32_t outer, int32_t di, int32_t hi, int32_t wi,
                                     int32_t dout, int32_t ho, int32_t wo, float fill, mri3d_stream_t stream) {
    MRI3D_REQUIRE(x && y && outer > 0 && di > 0 && hi > 0 && wi > 0 && dout > 0 && ho > 0 && wo > 0, MRI3D_EINVAL,
                  "crop_or_pad: bad arguments");
    MRI3D_REQUIRE((int64_t)dout * ho * wo < 0x7fffffffLL, MRI3D_ENOTSUP, "crop_or_pad: output volume too large");
    hipStream_t s = static_cast<hipStream_t>(stream);
    // offset of the output origin in input coordinates: crop -> +floor(diff/2); pad -> -floor(diff/2)
    auto off = [](int in, int out) { return in >= out ? (in - out) / 2 : -((out - in) / 2); };
    const int64_t total = (int64_t)outer * dout * ho * wo;
    hipLaunchKernelGGL(crop_or_pad_kernel, dim3(stream_grid(total, 256 * 4)), dim3(256), 0, s, x, y, outer, di, hi, wi, dout,
                       ho, wo, off(di, dout), off(hi, ho), off(wi, wo), fill);
    return check_launch("crop_or_pad");
}

extern "C" size_t mri3d_order_stats_workspace_bytes(void) {
    return (size_t)(kB0 + kMaxRanks * kB1 + kMaxRanks * kB2 + kMaxRanks) * sizeof(unsigned) +
           (size_t)kMaxRanks * sizeof(unsigned long long) + 64;
}

extern "C" int mri3d_order_stats_f32(const float* x, int64_t n, const int64_t* ranks, int32_t nranks, float* out,
                                     void* workspace, size_t ws_bytes, mri3d_stream_t stream) {
    MRI3D_REQUIRE(x && ranks && out && n > 0 && nranks > 0 && nranks <= kMaxRanks, MRI3D_EINVAL,
                  "order_stats: bad arguments (n=%lld, nranks=%d, max %d)", (long long)n, nranks, kMaxRanks);
    MRI3D_REQUIRE(n <= 0xffffffffLL, MRI3D_ENOTSUP, "order_stats: n must fit 32-bit counters");
    MRI3D_REQUIRE(workspace && ws_bytes >= mri3d_order_stats_workspace_bytes() &&
                      (reinterpret_cast<uintptr_t>(workspace) & 7) == 0,
                  MRI3D_EWORKSPACE, "order_stats: workspace %zu < %zu", ws_bytes, mri3d_order_stats_workspace_bytes());
    RankSet rs;
    rs.n = nranks;
    for (int i = 0; i < nranks; ++i) {
        MRI3D_REQUIRE(ranks[i] >= 0 && ranks[i] < n, MRI3D_EINVAL, "order_stats: rank %lld outside [0, %lld)",
                      (long long)ranks[i], (long long)n);
        rs.rank[i] = ranks[i];
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    unsigned long long* remaining = static_cast<unsigned long long*>(workspace);
    unsigned* hist0 = reinterpret_cast<unsigned*>(remaining + kMaxRanks);
    unsigned* hist1 = hist0 + kB0;
    unsigned* hist2 = hist1 + (size_t)kMaxRanks * kB1;
    unsigned* prefix = hist2 + (size_t)kMaxRanks * kB2;
    const size_t zero_bytes = (size_t)(kB0 + kMaxRanks * kB1 + kMaxRanks * kB2 + kMaxRanks) * sizeof(unsigned);
    if (hipMemsetAsync(hist0, 0, zero_bytes, s) != hipSuccess) {
        set_error("order_stats: hipMemsetAsync failed");
        return MRI3D_ELAUNCH;
    }
    const int grid = stream_grid(n, 256 * 8);
    hipLaunchKernelGGL(ostat_hist0_kernel, dim3(grid), dim3(256), 0, s, x, (long long)n, hist0);
    hipLaunchKernelGGL(ostat_select_kernel<0>, dim3(1), dim3(kMaxRanks), 0, s, rs, hist0, prefix, remaining, out);
    hipLaunchKernelGGL(ostat_hist_kernel<1>, dim3(grid), dim3(256), 0, s, x, (long long)n, prefix, hist1, nranks);
    hipLaunchKernelGGL(ostat_select_kernel<1>, dim3(1), dim3(kMaxRanks), 0, s, rs, hist1, prefix, remaining, out);
    hipLaunchKernelGGL(ostat_hist_kernel<2>, dim3(grid), dim3(256), 0, s, x, (long long)n, prefix, hist2, nranks);
    hipLaunchKernelGGL(ostat_select_kernel<2>, dim3(1), dim3(kMaxRanks), 0, s, rs, hist2, prefix, remaining, out);
    return check_launch("order_stats");
}

extern "C" int mri3d_piecewise_linear_f32(const float* x, float* y, int64_t n, const double* edges, const double* slope,
                                          const double* intercept, int32_t nseg, mri3d_stream_t stream) {
    MRI3D_REQUIRE(x && y && slope && intercept && n > 0 && nseg >= 1 && nseg <= 16 && (nseg == 1 || edges), MRI3D_EINVAL,
                  "piecewise_linear: bad arguments (nseg=%d, max 16)", nseg);
    PwlMap m;
    m.nseg = nseg;
    for (int i = 0; i < 16; ++i) {
        m.edge[i] = (i < nseg - 1) ? edges[i] : 0.0;
        m.slope[i] = (i < nseg) ? slope[i] : 0.0;
        m.icpt[i] = (i < nseg) ? intercept[i] : 0.0;
    }
    for (int i = 0; i + 2 < nseg; ++i)
        MRI3D_REQUIRE(!(edges[i + 1] < edges[i]), MRI3D_EINVAL, "piecewise_linear: edges must be non-decreasing");
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(pwl_kernel, dim3(stream_grid(n, 256 * 4)), dim3(256), 0, s, x, y, (long long)n, m);
    return check_launch("piecewise_linear");
}
