// conv_mfma3.hip — EXPERIMENT, not part of the shipped library (round 2): forward / data-gradient of Conv3d 3x3x3 / stride 1 / pad 1
// with 64-BYTE chunks.  Correct (the fuzz suite's 64-byte-chunk cases passed with it dispatched) but NOT faster than
// conv_mfma_fwd2_kernel, so it was taken out of the dispatch again.  Measured, 2 x 160x192x160, tools/conv_bench.py
// (profiles/r02_mfma3_experiment.txt):                       fwd2 kernel      this kernel (halo burst)   this kernel (spread loads)
//     fp32 48->16 forward                                    121.1 TFLOP/s    113.3                      118.8
//     fp32 16->16 forward                                    104.8            99.1                       102.3
//     bf16 48->16 forward                                    0.611 ms         0.834 ms                   0.765 ms
// What it taught: (1) a burst of halo loads in front of the tap loop stalls the weight ring for a whole HBM round trip per chunk
// (vmcnt is in order) — spreading them one per tap recovered 5 %; (2) ONE 512-thread workgroup per CU with two barriers per chunk
// has nobody to cover its store / start-up phases: the two independent 256-thread workgroups of fwd2 do that for each other, and
// that is worth more than the halved line fetches and the 3.6 % fewer MFMAs of this geometry; for bf16 a chunk is only 2.9 us of
// MFMAs, so the exposed phases weigh double.  A 64-byte geometry needs two co-resident workgroups per CU, i.e. <= 70 KB of LDS
// and <= 128 registers each — a different register plan than this one.
//
// Why a second geometry.  conv_mfma_fwd2_kernel consumes K in 32-byte chunks per voxel (8 fp32 / 16 bf16 channels): a chunk pass
// touches one whole 128-byte line per voxel for 32 useful bytes, so the 48 -> 16 layer fetches 12.0 GB (fp32) / 3.8 GB (bf16) per
// launch against 2.5 / 1.3 GB algorithmic (profiles/r02_hbm_traffic_48_16.json) — in bf16 that traffic, at the achievable HBM rate,
// IS the kernel's time (MFMA 34 % busy).  Here a chunk is 64 bytes per voxel (16 fp32 / 32 bf16 channels):
//   * half the chunk passes, each using half a line instead of a quarter: about half the fetched lines;
//   * an 8x8x16 output tile (halo 10x10x18: 1.76x instead of 2.11x re-read), one 512-thread workgroup per CU;
//   * one tap per MFMA k-group set: 27 taps, no padding tap (3.6 % fewer MFMAs than the two-taps-per-k-step pairing);
//   * half the per-chunk barriers and loop overheads.
// LDS: ONE halo tile (113 KB), stored as four QUARTER PLANES [q][voxel] of 16-byte pieces (q = which 16 bytes of the voxel's 64).
// A lane (voxel li, k-group kq) reads piece kq of its voxel: address = kq * PLANE + voxel * 16.  With PLANE a multiple of 256 bytes
// the 16 lanes of every hardware ds_read_b128 group ({0-3,12-15,20-27}, ...: eight lanes of k-group a, eight of a+1) hit the slots
// (v0 + li) mod 16 — all different, for ANY first voxel v0, so every tap / row offset is a compile-time immediate and no read
// needs a swizzle computation.  The next item's 15 pieces per lane are fetched into registers while the current chunk is
// multiplied and written between two barriers after it (as conv_mfma_wgrad6_kernel does); staging lanes are permuted so that
// 8 consecutive lanes write 8 consecutive voxels of one quarter plane (conflict-free ds_write_b128) while a wave still loads
// 16 whole voxels (coalesced).
// A wave owns one d-plane of the tile: 8 M-tiles (rows) x one N-tile, fragments of a (kd, kw) class shared by its three kh taps
// (10 row fragments for 24 MFMA groups), two fragment sets in registers.  Same operand order, K permutation, epilogue and fused
// BatchNorm statistics (STATS) as conv_mfma_fwd2_kernel.
#include "common.h"

namespace mri3d {

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace m3 {
constexpr int TD = 8, TH = 8, TW = 16;
constexpr int HD = TD + 2, HH = TH + 2, HW = TW + 2;
constexpr int NV = HD * HH * HW;              // 1800 halo voxels
constexpr int NVP = 1808;                     // padded so that a quarter plane is a multiple of 256 bytes (113 x 256)
constexpr int PLANE = NVP * 16;               // bytes
constexpr int NSTG = (NV * 4 + 511) / 512;    // 15 pieces per lane
constexpr int LDS_TILE = 4 * PLANE;           // 115712 bytes
constexpr int LDS_OFFS = 16 * 512 * 4;        // per-lane staging offsets (16 dwords x 512 lanes): registers are the scarce resource
}  // namespace m3

// sum over the 16 lanes of a DPP row (lanes with equal lane >> 4)
__device__ __forceinline__ float m3_row_sum16(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));
    return v;
}

// 16-byte load through an explicitly GLOBAL pointer (global_load_dwordx4: vmcnt only)
template <typename T>
__device__ __forceinline__ float4 m3_ldg16(const T* p) {
    typedef float gf32x4 __attribute__((ext_vector_type(4)));
    const gf32x4 v = *(const __attribute__((address_space(1))) gf32x4*)p;
    return make_float4(v[0], v[1], v[2], v[3]);
}

// ------------------------------------------------------------------ weight packing
// Wp[chunk][tap][nt][lane][s]:  nc = nt*16 + (lane & 15),  kc = chunk*CK + PE*(lane >> 4) + s   (PE = 4 fp32 / 8 bf16 per piece)
//   forward: W'(tap,kc,nc) = W[nc][kc][tap];  dgrad: W'(tap,kc,nc) = W[kc][nc][26 - tap]
template <typename TW_>
__global__ void pack_w_mfma3_kernel(const float* __restrict__ w, TW_* __restrict__ wp, int Co, int Ci, int dgrad, int NTT,
                                    int nchunks) {
    constexpr int PE = 16 / sizeof(TW_);
    const int per = 64 * PE;                                  // elements per (chunk, tap, nt): 1 KiB
    const int total = nchunks * 27 * NTT * per;
    const int Kc = dgrad ? Co : Ci, Nc = dgrad ? Ci : Co;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int s = i % PE, lane = (i / PE) & 63;
        int t = i / per;
        const int nt = t % NTT;
        t /= NTT;
        const int tap = t % 27, chunk = t / 27;
        const int nc = nt * 16 + (lane & 15);
        const int kc = chunk * (4 * PE) + PE * (lane >> 4) + s;
        float v = 0.f;
        if (nc < Nc && kc < Kc) v = dgrad ? w[((size_t)kc * Ci + nc) * 27 + (26 - tap)] : w[((size_t)nc * Ci + kc) * 27 + tap];
        wp[i] = (TW_)v;
    }
}

template <typename T, bool STATS>
__global__ void __launch_bounds__(512, 2)
conv_mfma3_kernel(const T* __restrict__ x, const float* __restrict__ wp, const float* __restrict__ bias, T* __restrict__ y,
                  int N, int D, int H, int W, int Kc, int x_ld, int Nc, int y_ld, int NTT, int tilesD, int tilesH, int tilesW,
                  int ntiles, double* __restrict__ stat_part) {
    using namespace m3;
    constexpr bool kBf16 = sizeof(T) == 2;
    constexpr int PE = 16 / sizeof(T);        // channels per 16-byte piece
    constexpr int CK = 4 * PE;                // channels per 64-byte chunk (16 fp32 / 32 bf16)
    extern __shared__ __attribute__((aligned(16))) char lds3[];

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;   // wv = the wave's output d-plane of the tile
    const int li = lane & 15, kq = lane >> 4;
    const int nchunks = (Kc + CK - 1) / CK;
    // Tile -> workgroup map (speed only), as in conv_mfma_fwd2_kernel: XCD k = b & 7 owns a contiguous range of (tile, N-block)
    // units and its workgroups take them round-robin, so concurrently processed tiles are neighbours inside one XCD's L2.
    const int NX = gridDim.x < 8 ? (int)gridDim.x : 8;
    const int xcd = blockIdx.x % NX, wslot = blockIdx.x / NX;
    const int wper = ((int)gridDim.x - xcd + NX - 1) / NX;
    const int r_lo = (int)(((int64_t)ntiles * xcd) / NX), r_hi = (int)(((int64_t)ntiles * (xcd + 1)) / NX);
    const int my_tiles = (r_hi - r_lo - wslot + wper - 1) / wper;
    const int nitems = (r_lo + wslot < r_hi ? my_tiles : 0) * nchunks;
    unsigned* const offs_lds = reinterpret_cast<unsigned*>(lds3 + LDS_TILE) + tid * 16;   // this lane's 15 piece offsets
    double* const stat_lds = reinterpret_cast<double*>(lds3 + LDS_TILE + LDS_OFFS);   // [8 waves][NTT * 16 channels][2]
    if constexpr (STATS) {
        if (nitems <= 0) {
            for (int i = tid; i < Nc * 2; i += 512) stat_part[(size_t)blockIdx.x * Nc * 2 + i] = 0.0;
            return;
        }
        for (int i = tid; i < 8 * NTT * 16 * 2; i += 512) stat_lds[i] = 0.0;   // visible after the first barrier pair
    }
    if (nitems <= 0) return;

    // staging roles: piece p = j*512 + tid covers halo voxel v = j*128 + vlane, quarter q (32 consecutive pieces = 8 voxels x 4
    // quarters; lanes 0..7 of such a group take the 8 voxels of quarter 0, lanes 8..15 those of quarter 1, ...)
    const int vlane = (tid >> 5) * 8 + (tid & 7), q = (tid >> 3) & 3;
    // element offset of piece j from the halo origin voxel (d0-1, h0-1, w0-1): kept in LDS (one 64-byte row per lane, read back as
    // four ds_read_b128 per chunk) — 15 registers the fragment / prefetch sets need more
#pragma unroll
    for (int j = 0; j < NSTG; ++j) {
        const int v = j * 128 + vlane;
        const int vv = v < NV ? v : 0;
        const int wx = vv % HW, t2 = vv / HW;
        offs_lds[j] = (unsigned)((((t2 / HH) * H + t2 % HH) * W + wx) * x_ld + PE * q);
    }
    offs_lds[15] = 0u;
    const unsigned forig = (unsigned)(((H + 1) * W + 1) * x_ld);   // the tile's first output voxel: always inside the tensor
    char* const lds_w = lds3 + q * PLANE + vlane * 16;              // + j * 2048: where piece j goes
    const int last_v = (NSTG - 1) * 128 + vlane;                    // only the last piece index can fall outside the tile
    char* const lds_w_last = last_v < NV ? lds_w + (NSTG - 1) * 2048 : lds3 + q * PLANE + (NV + (tid & 7)) * 16;   // padding slots

    struct Item { int n, d0, h0, w0, nt0, ch; };
    auto decode = [&](int it) -> Item {
        Item r;
        int tile = r_lo + wslot + (it / nchunks) * wper;
        r.ch = it % nchunks;
        r.nt0 = tile % NTT;          // one N-tile per pass: N-blocks fastest, so the passes of a tile follow each other
        tile /= NTT;
        r.w0 = (tile % tilesW) * TW;
        tile /= tilesW;
        r.h0 = (tile % tilesH) * TH;
        tile /= tilesH;
        r.d0 = (tile % tilesD) * TD;
        r.n = tile / tilesD;
        return r;
    };
    auto halo_origin = [&](const Item& it) -> const T* {   // wave-uniform; only dereferenced at in-volume offsets
        return x + (((((int64_t)it.n * D + it.d0 - 1) * H + it.h0 - 1) * W + it.w0 - 1) * x_ld + it.ch * CK);
    };

    float4 pre[NSTG];       // the next item's pieces
    unsigned okmask = 0;    // which of them lie inside the tensor (the others are zero-filled when written)
    // which pieces of an item lie inside the tensor (VALU only; wave-uniform shortcut for interior tiles)
    auto item_mask = [&](const Item& it) -> unsigned {
        const bool interior = it.d0 >= 1 && it.d0 + TD < D && it.h0 >= 1 && it.h0 + TH < H && it.w0 >= 1 && it.w0 + TW < W &&
                              (it.ch + 1) * CK <= Kc;   // scalars
        unsigned ok = (1u << NSTG) - 1u;
        if (last_v >= NV) ok &= ~(1u << (NSTG - 1));
        if (!interior) {
            const bool qok = it.ch * CK + PE * q < Kc;
#pragma unroll
            for (int j = 0; j < NSTG; ++j) {
                const int v = j * 128 + vlane;
                const int vv = v < NV ? v : 0;
                const int wx = vv % HW, t2 = vv / HW;
                const int gd = it.d0 - 1 + t2 / HH, gh = it.h0 - 1 + t2 % HH, gw = it.w0 - 1 + wx;
                const bool in = qok && (unsigned)gd < (unsigned)D && (unsigned)gh < (unsigned)H && (unsigned)gw < (unsigned)W;
                if (!in) ok &= ~(1u << j);
            }
        }
        return ok;
    };
    // piece j of an item: unconditional load at a clamped offset (straight-line code: counted vmcnt waits stay exact)
    auto load_piece = [&](const T* fb, unsigned ok, int j, unsigned off) -> float4 {
        return m3_ldg16(fb + (((ok >> j) & 1u) ? off : forig));
    };
    const unsigned allok = last_v < NV ? (1u << NSTG) - 1u : ((1u << NSTG) - 1u) & ~(1u << (NSTG - 1));
    auto store_item = [&]() {
        if (okmask == allok) {   // every real piece is inside the tensor (the unused last piece goes to a padding slot as it is)
#pragma unroll
            for (int j = 0; j < NSTG - 1; ++j) *reinterpret_cast<float4*>(lds_w + j * 2048) = pre[j];
            *reinterpret_cast<float4*>(lds_w_last) = pre[NSTG - 1];
        } else {
#pragma unroll
            for (int j = 0; j < NSTG; ++j) {
                const bool in = (okmask >> j) & 1u;
                float4 v2;
                v2.x = in ? pre[j].x : 0.f;
                v2.y = in ? pre[j].y : 0.f;
                v2.z = in ? pre[j].z : 0.f;
                v2.w = in ? pre[j].w : 0.f;
                *reinterpret_cast<float4*>(j < NSTG - 1 ? lds_w + j * 2048 : lds_w_last) = v2;
            }
        }
    };

    f32x4 acc[TH];
#pragma unroll
    for (int m = 0; m < TH; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};

    // A-fragment base: piece kq of voxel (plane wv + kd, row i, w = li + kw); every (kd, i, kw) is an immediate offset
    const char* const fbase = lds3 + kq * PLANE + ((wv * HH) * HW + li) * 16;
    auto frag = [&](int kd, int i, int kw) -> f32x4 {
        return *reinterpret_cast<const f32x4*>(fbase + ((kd * HH + i) * HW + kw) * 16);
    };

    Item cur = decode(0);
    okmask = item_mask(cur);
    {
        const T* fb0 = halo_origin(cur);
#pragma unroll
        for (int j = 0; j < NSTG; ++j) pre[j] = load_piece(fb0, okmask, j, offs_lds[j]);
    }
    for (int it = 0; it < nitems; ++it) {
        __syncthreads();   // every wave is done reading the previous chunk
        store_item();
        __syncthreads();
        const bool has_next = it + 1 < nitems;
        Item nxt = cur;
        if (has_next) {
            if (cur.ch + 1 < nchunks) nxt.ch = cur.ch + 1;
            else nxt = decode(it + 1);
        }
        // weights of the first three taps BEFORE the staging burst: their waits must not include the 15 halo loads
        const float* wt = wp + ((size_t)cur.ch * 27 * NTT + cur.nt0) * 256 + lane * 4;
        const size_t wstep = (size_t)NTT * 256;
        f32x4 bq[3];
        // execution order e = class * 3 + kh, class = kd * 3 + kw  ->  tap index (kd * 3 + kh) * 3 + kw
        auto tap_of = [](int e) { return ((e / 9) * 3 + e % 3) * 3 + (e / 3) % 3; };
        bq[0] = *reinterpret_cast<const f32x4*>(wt + (size_t)tap_of(0) * wstep);
        bq[1] = *reinterpret_cast<const f32x4*>(wt + (size_t)tap_of(1) * wstep);
        bq[2] = *reinterpret_cast<const f32x4*>(wt + (size_t)tap_of(2) * wstep);
        // The next item's 15 pieces are fetched ONE PER TAP (taps 0..14), each right before its tap's MFMAs: a weight load
        // issued later can only be waited for after every older load has landed (vmcnt is in order), so a burst of 15 halo loads
        // in front of the tap loop stalls the weight ring for a whole HBM round trip per chunk (measured: 113 instead of 121
        // TFLOP/s).  Spread out, the weights of tap e + 3 only queue behind pieces issued three taps before they are needed.
        const Item stg = has_next ? nxt : cur;             // unconditional: the last chunk re-reads its own item
        const T* fbn = halo_origin(stg);
        const unsigned nmask = item_mask(stg);
        unsigned off_next = offs_lds[0];

        f32x4 fr[2][TH + 2];
#pragma unroll
        for (int i = 0; i < TH + 2; ++i) fr[0][i] = frag(0, i, 0);
#pragma unroll
        for (int e = 0; e < 27; ++e) {
            const int cls = e / 3, kh = e % 3;
            if (e < NSTG) {
                pre[e] = load_piece(fbn, nmask, e, off_next);
                if (e + 1 < NSTG) off_next = offs_lds[e + 1];
            }
            if (cls + 1 < 9) {   // this tap's share of the next class's fragments
                const int ncls = cls + 1, nkd = ncls / 3, nkw = ncls % 3;
#pragma unroll
                for (int i = 0; i < TH + 2; ++i)
                    if (i >= kh * (TH + 2) / 3 && i < (kh + 1) * (TH + 2) / 3) fr[ncls & 1][i] = frag(nkd, i, nkw);
            }
            __builtin_amdgcn_sched_barrier(0);   // keep the prefetches above this tap's MFMAs
            if constexpr (kBf16) {
#pragma unroll
                for (int m = 0; m < TH; ++m)
                    acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, bq[e % 3]),
                                                                     __builtin_bit_cast(bf16x8_t, fr[cls & 1][m + kh]), acc[m], 0, 0, 0);
            } else {
#pragma unroll
                for (int m = 0; m < TH; m += 2)
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(bq[e % 3][s], fr[cls & 1][m + kh][s], acc[m], 0, 0, 0);
                        acc[m + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(bq[e % 3][s], fr[cls & 1][m + 1 + kh][s], acc[m + 1], 0, 0, 0);
                    }
            }
            // the ring slot this tap used is free again: weights of tap e + 3 (two taps of MFMAs cover the load)
            if (e + 3 < 27) bq[e % 3] = *reinterpret_cast<const f32x4*>(wt + (size_t)tap_of(e + 3) * wstep);
        }

        if (cur.ch == nchunks - 1) {
            // epilogue: lane holds channels 4*kq..4*kq+3 of voxel li of every row
            const int od = cur.d0 + wv, ow = cur.w0 + li;
            const int co = cur.nt0 * 16 + 4 * kq;
            if (od < D && ow < W && co < Nc) {
                const bool vec = (co + 3 < Nc) && ((y_ld & 3) == 0);
                float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
                if (bias) {
                    bv.x = bias[co];
                    if (co + 1 < Nc) bv.y = bias[co + 1];
                    if (co + 2 < Nc) bv.z = bias[co + 2];
                    if (co + 3 < Nc) bv.w = bias[co + 3];
                }
#pragma unroll
                for (int m = 0; m < TH; ++m) {
                    const int oh = cur.h0 + m;
                    if (oh < H) {
                        T* yp = y + ((((int64_t)cur.n * D + od) * H + oh) * W + ow) * y_ld + co;
                        const f32x4 a = acc[m];
                        if (vec) {
                            stf4(yp, make_float4(a[0] + bv.x, a[1] + bv.y, a[2] + bv.z, a[3] + bv.w));
                        } else {
                            stf(yp, a[0] + bv.x);
                            if (co + 1 < Nc) stf(yp + 1, a[1] + bv.y);
                            if (co + 2 < Nc) stf(yp + 2, a[2] + bv.z);
                            if (co + 3 < Nc) stf(yp + 3, a[3] + bv.w);
                        }
                    }
                }
            }
            if constexpr (STATS) {
                const bool vok = od < D && ow < W;
                float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int m = 0; m < TH; ++m) {
                    const bool ok = vok && cur.h0 + m < H;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float a = ok ? acc[m][r] : 0.f;
                        s1[r] += a;
                        s2[r] += a * a;
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    s1[r] = m3_row_sum16(s1[r]);
                    s2[r] = m3_row_sum16(s2[r]);
                }
                if (li == 0) {
                    double* slot = stat_lds + ((size_t)wv * NTT * 16 + cur.nt0 * 16 + 4 * kq) * 2;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        slot[2 * r] += (double)s1[r];
                        slot[2 * r + 1] += (double)s2[r];
                    }
                }
            }
#pragma unroll
            for (int m = 0; m < TH; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        okmask = nmask;
        cur = nxt;
    }
    if constexpr (STATS) {
        __syncthreads();
        for (int i = tid; i < Nc * 2; i += 512) {
            double v = 0.0;
#pragma unroll
            for (int w8 = 0; w8 < 8; ++w8) v += stat_lds[(size_t)w8 * NTT * 16 * 2 + i];
            stat_part[(size_t)blockIdx.x * Nc * 2 + i] = v;
        }
    }
}

// ------------------------------------------------------------------ host side
struct Mfma3Plan {
    int NTT, nchunks, tilesD, tilesH, tilesW, ntiles, grid;
    size_t wp_bytes, smem, stat_smem;
};

// Which geometries take this kernel: 3x3x3 / stride 1 / pad 1, an ODD number of 16-channel N-tiles (one N-tile per pass: Cout 16
// or 48 forward, Cin 16 or 48 data gradient; even counts keep the two-N-tile kernel), K a whole number of 64-byte chunks or at
// least 1.5 of them (a half-empty chunk multiplies zeros: acceptable for bf16's 48 channels, not for 8 or 16 of them), and enough
// tiles to fill the chip.
bool conv_mfma3_plan(const Mri3dConvGeom& g, bool dgrad, Mfma3Plan& p) {
    using namespace m3;
    if (!(g.kd == 3 && g.kh == 3 && g.kw == 3 && g.sd == 1 && g.sh == 1 && g.sw == 1 && g.pd == 1 && g.ph == 1 &&
          g.pw == 1 && g.dd == 1 && g.dh == 1 && g.dw == 1))
        return false;
    const bool bf = g.dtype == MRI3D_BF16;
    const int Kc = dgrad ? g.co : g.ci, Nc = dgrad ? g.ci : g.co;
    const int in_ld = dgrad ? g.y_ld : g.x_ld;
    const int PE = bf ? 8 : 4, CK = 4 * PE;
    if (Kc % PE != 0 || in_ld % PE != 0) return false;
    if (Kc % CK != 0 && Kc < CK + CK / 2) return false;
    if (Nc < 8) return false;
    p.NTT = cdiv(Nc, 16);
    if (p.NTT % 2 == 0 || p.NTT > 5) return false;   // LDS: tile 113 KB + offsets 32 KB + statistics 2 KB per N-tile
    p.nchunks = cdiv(Kc, CK);
    p.tilesD = cdiv(g.di, TD);
    p.tilesH = cdiv(g.hi, TH);
    p.tilesW = cdiv(g.wi, TW);
    const int64_t nt = (int64_t)g.n * p.tilesD * p.tilesH * p.tilesW * p.NTT;
    if (nt > 0x7fffffff || nt < 256) return false;   // fewer units than CUs: the tiled / small kernels of conv_mfma.hip
    p.ntiles = (int)nt;
    p.grid = (int)std::min<int64_t>(nt, 256);         // one resident 512-thread workgroup per CU
    p.wp_bytes = (size_t)p.nchunks * 27 * p.NTT * 1024;
    p.smem = LDS_TILE + LDS_OFFS;
    p.stat_smem = (size_t)8 * p.NTT * 16 * 2 * sizeof(double);
    return true;
}

size_t conv_mfma3_workspace_bytes(const Mri3dConvGeom& g, bool dgrad) {
    Mfma3Plan p;
    return conv_mfma3_plan(g, dgrad, p) ? p.wp_bytes : 0;
}

int conv_mfma3_stat_blocks(const Mri3dConvGeom& g) {
    Mfma3Plan p;
    return conv_mfma3_plan(g, false, p) ? p.grid : 0;
}

int conv_mfma3_run(const Mri3dConvGeom& g, bool dgrad, const void* in_v, const float* w, const float* bias, void* out_v, void* ws,
                   size_t ws_bytes, hipStream_t s, double* stat_part) {
    Mfma3Plan p;
    MRI3D_REQUIRE(conv_mfma3_plan(g, dgrad, p), MRI3D_ENOTSUP, "conv3d(mfma3): unsupported geometry");
    MRI3D_REQUIRE(ws && ws_bytes >= p.wp_bytes, MRI3D_EWORKSPACE, "conv3d(mfma3): workspace %zu < %zu", ws_bytes, p.wp_bytes);
    MRI3D_REQUIRE(((reinterpret_cast<uintptr_t>(in_v) | reinterpret_cast<uintptr_t>(out_v) | reinterpret_cast<uintptr_t>(ws)) & 15) == 0,
                  MRI3D_EINVAL, "conv3d(mfma3): input/output/workspace must be 16-byte aligned");
    const int Kc = dgrad ? g.co : g.ci, Nc = dgrad ? g.ci : g.co;
    const int in_ld = dgrad ? g.y_ld : g.x_ld, out_ld = dgrad ? g.x_ld : g.y_ld;
    const int pk = (int)std::min<size_t>(cdiv((int)(p.wp_bytes / 4), 256), 2048);
    if (g.dtype == MRI3D_BF16)
        hipLaunchKernelGGL(pack_w_mfma3_kernel<bf16_t>, dim3(pk), dim3(256), 0, s, w, static_cast<bf16_t*>(ws), g.co, g.ci,
                           dgrad ? 1 : 0, p.NTT, p.nchunks);
    else
        hipLaunchKernelGGL(pack_w_mfma3_kernel<float>, dim3(pk), dim3(256), 0, s, w, static_cast<float*>(ws), g.co, g.ci,
                           dgrad ? 1 : 0, p.NTT, p.nchunks);
    const size_t smem = p.smem + (stat_part ? p.stat_smem : 0);
    constexpr int kMaxSmem = m3::LDS_TILE + m3::LDS_OFFS + 8 * 5 * 16 * 2 * 8;
#define MRI3D_M3_CASE(STv)                                                                                            \
    if ((stat_part != nullptr) == STv) {                                                                              \
        auto kern = conv_mfma3_kernel<T, STv>;                                                                        \
        static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                       \
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, kMaxSmem);     \
        (void)attr;                                                                                                   \
        hipLaunchKernelGGL(kern, dim3(p.grid), dim3(512), smem, s, (const T*)in_v, (const float*)ws, bias, (T*)out_v, g.n, \
                           g.di, g.hi, g.wi, Kc, in_ld, Nc, out_ld, p.NTT, p.tilesD, p.tilesH, p.tilesW, p.ntiles,       \
                           stat_part);                                                                                \
    }
    MRI3D_DISPATCH_DTYPE(g.dtype, T, {
        MRI3D_M3_CASE(false)
        MRI3D_M3_CASE(true)
    });
#undef MRI3D_M3_CASE
    return check_launch(dgrad ? "conv3d_dgrad(mfma3)" : "conv3d_fwd(mfma3)");
}

}  // namespace mri3d
