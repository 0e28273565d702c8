// surface.hip — average surface distance of two binary masks on the device: what validate_dsc_asd asks
// of compute_average_surface_distance(compute_surface_distances(gt, pred, spacing_mm=(1,1,1))) (segmentation/routine.py:205-214,
// segmentation/metrics.py:25-207; 5-7 s per volume on the CPU, results_validation.ipynb:267).
//
//   1. neighbour codes: the 2x2x2 local binary pattern at every voxel CORNER (a (D+1)x(H+1)x(W+1) grid), weights
//      128,64,32,16,8,4,2,1 as in metrics.py:118-125; a corner is a surface element iff its code is neither 0 nor 255.
//   2. exact Euclidean distance transform to the nearest surface element of the OTHER mask: three separable passes over
//      squared integer distances (min-plus along W, H, D — brute force per line, 161..193 candidates, all in cache); integers,
//      so the squared distance is exact and sqrt() in double equals scipy's distance_transform_edt for unit spacing.
//   3. area-weighted sums: sum(dist * area[code]) and sum(area[code]) over the surface elements of each mask, double
//      block partials combined in a fixed order.  The host divides.
// The bounding-box crop of the reference only shrinks the work; distances to the nearest surface element are unchanged.
#include "common.h"
#include <algorithm>

namespace mri3d {

constexpr int kSdInf = 0x3f000000;   // "no surface element on this line" (squared distances stay < 2^18)
constexpr int kSdBlocks = 512;

struct AreaTable { double a[256]; };

__device__ __forceinline__ bool is_border(unsigned char c) { return c != 0 && c != 255; }

// codes for both masks: corner (i,j,k) looks at voxels (i-1..i, j-1..j, k-1..k)
__global__ void __launch_bounds__(256)
sd_codes_kernel(const uint8_t* __restrict__ gt, const uint8_t* __restrict__ pred, uint8_t* __restrict__ cg,
                uint8_t* __restrict__ cp, int D, int H, int W) {
    const int Hc = H + 1, Wc = W + 1;
    const long long total = (long long)(D + 1) * Hc * Wc;
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (long long)gridDim.x * blockDim.x) {
        const int k = (int)(p % Wc);
        const long long t = p / Wc;
        const int j = (int)(t % Hc), i = (int)(t / Hc);
        unsigned a = 0, b = 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int vi = i - 1 + (q >> 2), vj = j - 1 + ((q >> 1) & 1), vk = k - 1 + (q & 1);
            if ((unsigned)vi < (unsigned)D && (unsigned)vj < (unsigned)H && (unsigned)vk < (unsigned)W) {
                const long long v = ((long long)vi * H + vj) * W + vk;
                const unsigned wgt = 128u >> q;
                a += gt[v] ? wgt : 0u;
                b += pred[v] ? wgt : 0u;
            }
        }
        cg[p] = (uint8_t)a;
        cp[p] = (uint8_t)b;
    }
}

// pass along W: squared distance to the nearest surface element in the same (i, j) row
__global__ void __launch_bounds__(256)
sd_edt_w_kernel(const uint8_t* __restrict__ code, int* __restrict__ f, int Dc, int Hc, int Wc) {
    const long long total = (long long)Dc * Hc * Wc;
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (long long)gridDim.x * blockDim.x) {
        const int k = (int)(p % Wc);
        const uint8_t* row = code + (p - k);
        int best = kSdInf;
        for (int q = 0; q < Wc; ++q)
            if (is_border(row[q])) { const int d = k - q; best = min(best, d * d); }
        f[p] = best;
    }
}

// min-plus pass along an axis with element stride `st` and length `len`: out[x] = min_q in[q] + (x - q)^2
__global__ void __launch_bounds__(256)
sd_edt_axis_kernel(const int* __restrict__ in, int* __restrict__ out, long long total, long long st, int len) {
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (long long)gridDim.x * blockDim.x) {
        const int xpos = (int)((p / st) % len);
        const int* line = in + (p - (long long)xpos * st);
        int best = kSdInf;
        for (int q = 0; q < len; ++q) {
            const int d = xpos - q;
            best = min(best, line[(long long)q * st] + d * d);   // kSdInf + d*d cannot overflow
        }
        out[p] = best;
    }
}

// part[b] = (sum dist*area, sum area) over the surface elements of `code_self`, distances from `f_other`
__global__ void __launch_bounds__(256)
sd_reduce_kernel(const uint8_t* __restrict__ code_self, const int* __restrict__ f_other, long long total, AreaTable tab,
                 double* __restrict__ part) {
    __shared__ double red[4][2];
    double sda = 0.0, sa = 0.0;
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (long long)gridDim.x * blockDim.x) {
        const uint8_t c = code_self[p];
        if (is_border(c)) {
            const int f2 = f_other[p];
            const double dist = f2 >= kSdInf ? (double)INFINITY : sqrt((double)f2);
            const double ar = tab.a[c];
            sda += dist * ar;
            sa += ar;
        }
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    sda = wave_sum_d(sda);
    sa = wave_sum_d(sa);
    if (lane == 0) { red[wave][0] = sda; red[wave][1] = sa; }
    __syncthreads();
    if (threadIdx.x < 2)
        part[(size_t)blockIdx.x * 2 + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// append (squared distance to the other surface, neighbour code) of every surface element of `code_self`; the order of the
// list is arbitrary (atomic cursor) — the host sorts it the way the reference does
__global__ void __launch_bounds__(256)
sd_collect_kernel(const uint8_t* __restrict__ code_self, const int* __restrict__ f_other, long long total,
                  int* __restrict__ out_d2, uint8_t* __restrict__ out_code, unsigned long long* __restrict__ cursor,
                  long long capacity) {
    const int lane = threadIdx.x & 63;
    for (long long base = (long long)blockIdx.x * blockDim.x; base < total; base += (long long)gridDim.x * blockDim.x) {
        const long long p = base + threadIdx.x;
        const uint8_t c = p < total ? code_self[p] : (uint8_t)0;
        const bool hit = is_border(c);
        const unsigned long long m = __ballot(hit);
        if (m == 0ull) continue;   // wave-uniform
        unsigned long long start = 0;
        if (lane == __ffsll((long long)m) - 1) start = atomicAdd(cursor, (unsigned long long)__popcll(m));
        start = __shfl(start, __ffsll((long long)m) - 1, 64);
        if (hit) {
            const long long idx = (long long)start + __popcll(m & ((1ull << lane) - 1ull));
            if (idx < capacity) {
                out_d2[idx] = f_other[p];
                out_code[idx] = c;
            }
        }
    }
}

__global__ void sd_finalize_kernel(const double* __restrict__ part, int nblk, double* __restrict__ out2) {
    if (threadIdx.x >= 2) return;
    double s = 0.0;
    for (int b = 0; b < nblk; ++b) s += part[(size_t)b * 2 + threadIdx.x];
    out2[threadIdx.x] = s;
}

}  // namespace mri3d

using namespace mri3d;

static inline size_t sd_corners(int d, int h, int w) { return (size_t)(d + 1) * (h + 1) * (w + 1); }

extern "C" size_t mri3d_surface_distance_workspace_bytes(int32_t d, int32_t h, int32_t w) {
    if (d <= 0 || h <= 0 || w <= 0) return 0;
    const size_t nc = sd_corners(d, h, w);
    return align_up(nc, 256) * 2 + nc * sizeof(int) * 3 + (size_t)kSdBlocks * 2 * sizeof(double) + 1024;
}

extern "C" int mri3d_surface_distance(const uint8_t* gt, const uint8_t* pred, int32_t d, int32_t h, int32_t w,
                                      const double* area_table, double* sums, void* workspace, size_t ws_bytes,
                                      mri3d_stream_t stream) {
    MRI3D_REQUIRE(gt && pred && area_table && sums && d > 0 && h > 0 && w > 0, MRI3D_EINVAL, "surface_distance: bad arguments");
    MRI3D_REQUIRE((int64_t)(d + 1) * (h + 1) * (w + 1) < 0x7fffffffLL && d < 32768 && h < 32768 && w < 32768, MRI3D_ENOTSUP,
                  "surface_distance: volume too large");
    MRI3D_REQUIRE(workspace && ws_bytes >= mri3d_surface_distance_workspace_bytes(d, h, w) &&
                      (reinterpret_cast<uintptr_t>(workspace) & 7) == 0,
                  MRI3D_EWORKSPACE, "surface_distance: workspace %zu < %zu", ws_bytes,
                  mri3d_surface_distance_workspace_bytes(d, h, w));
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int Dc = d + 1, Hc = h + 1, Wc = w + 1;
    const long long nc = (long long)Dc * Hc * Wc;
    char* base = static_cast<char*>(workspace);
    double* part = reinterpret_cast<double*>(base);
    base += align_up((size_t)kSdBlocks * 2 * sizeof(double), 256);
    int* fa = reinterpret_cast<int*>(base);
    int* fb = fa + nc;
    int* fc = fb + nc;
    uint8_t* cg = reinterpret_cast<uint8_t*>(fc + nc);
    uint8_t* cp = cg + align_up((size_t)nc, 256);
    AreaTable tab;
    for (int i = 0; i < 256; ++i) tab.a[i] = area_table[i];
    const int grid = stream_grid(nc, 256);
    const int nblk = std::min(kSdBlocks, grid);
    hipLaunchKernelGGL(sd_codes_kernel, dim3(grid), dim3(256), 0, s, gt, pred, cg, cp, d, h, w);
    // direction 0: distances from gt surface elements to the pred surface (EDT of pred); direction 1: the converse
    for (int dir = 0; dir < 2; ++dir) {
        const uint8_t* other = dir == 0 ? cp : cg;
        const uint8_t* self = dir == 0 ? cg : cp;
        hipLaunchKernelGGL(sd_edt_w_kernel, dim3(grid), dim3(256), 0, s, other, fa, Dc, Hc, Wc);
        hipLaunchKernelGGL(sd_edt_axis_kernel, dim3(grid), dim3(256), 0, s, fa, fb, nc, (long long)Wc, Hc);
        hipLaunchKernelGGL(sd_edt_axis_kernel, dim3(grid), dim3(256), 0, s, fb, fc, nc, (long long)Wc * Hc, Dc);
        hipLaunchKernelGGL(sd_reduce_kernel, dim3(nblk), dim3(256), 0, s, self, fc, nc, tab, part);
        hipLaunchKernelGGL(sd_finalize_kernel, dim3(1), dim3(64), 0, s, part, nblk, sums + 2 * dir);
    }
    return check_launch("surface_distance");
}

/* Surface-element lists for the order-dependent metrics of segmentation/metrics.py:208-310 (robust Hausdorff, surface overlap /
 * surface Dice at a tolerance): for each direction every surface element's squared distance (exact integer; >= 0x3f000000 =
 * the other mask has no surface) and neighbour code, in arbitrary order; counts[dir] = number of elements (device uint64). */
extern "C" int mri3d_surface_elements(const uint8_t* gt, const uint8_t* pred, int32_t d, int32_t h, int32_t w, int32_t* d2_gt,
                                      uint8_t* code_gt, int32_t* d2_pred, uint8_t* code_pred, int64_t capacity,
                                      uint64_t* counts, void* workspace, size_t ws_bytes, mri3d_stream_t stream) {
    MRI3D_REQUIRE(gt && pred && d2_gt && code_gt && d2_pred && code_pred && counts && capacity > 0 && d > 0 && h > 0 && w > 0,
                  MRI3D_EINVAL, "surface_elements: bad arguments");
    MRI3D_REQUIRE((int64_t)(d + 1) * (h + 1) * (w + 1) < 0x7fffffffLL && d < 32768 && h < 32768 && w < 32768, MRI3D_ENOTSUP,
                  "surface_elements: volume too large");
    MRI3D_REQUIRE(workspace && ws_bytes >= mri3d_surface_distance_workspace_bytes(d, h, w) &&
                      (reinterpret_cast<uintptr_t>(workspace) & 7) == 0,
                  MRI3D_EWORKSPACE, "surface_elements: workspace %zu < %zu", ws_bytes,
                  mri3d_surface_distance_workspace_bytes(d, h, w));
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int Dc = d + 1, Hc = h + 1, Wc = w + 1;
    const long long nc = (long long)Dc * Hc * Wc;
    char* base = static_cast<char*>(workspace);
    base += align_up((size_t)kSdBlocks * 2 * sizeof(double), 256);
    int* fa = reinterpret_cast<int*>(base);
    int* fb = fa + nc;
    int* fc = fb + nc;
    uint8_t* cg = reinterpret_cast<uint8_t*>(fc + nc);
    uint8_t* cp = cg + align_up((size_t)nc, 256);
    const int grid = stream_grid(nc, 256);
    if (hipMemsetAsync(counts, 0, 2 * sizeof(uint64_t), s) != hipSuccess) {
        set_error("surface_elements: hipMemsetAsync failed");
        return MRI3D_ELAUNCH;
    }
    hipLaunchKernelGGL(sd_codes_kernel, dim3(grid), dim3(256), 0, s, gt, pred, cg, cp, d, h, w);
    for (int dir = 0; dir < 2; ++dir) {
        const uint8_t* other = dir == 0 ? cp : cg;
        const uint8_t* self = dir == 0 ? cg : cp;
        hipLaunchKernelGGL(sd_edt_w_kernel, dim3(grid), dim3(256), 0, s, other, fa, Dc, Hc, Wc);
        hipLaunchKernelGGL(sd_edt_axis_kernel, dim3(grid), dim3(256), 0, s, fa, fb, nc, (long long)Wc, Hc);
        hipLaunchKernelGGL(sd_edt_axis_kernel, dim3(grid), dim3(256), 0, s, fb, fc, nc, (long long)Wc * Hc, Dc);
        hipLaunchKernelGGL(sd_collect_kernel, dim3(grid), dim3(256), 0, s, self, fc, nc, dir == 0 ? d2_gt : d2_pred,
                           dir == 0 ? code_gt : code_pred, reinterpret_cast<unsigned long long*>(counts) + dir,
                           (long long)capacity);
    }
    return check_launch("surface_elements");
}
