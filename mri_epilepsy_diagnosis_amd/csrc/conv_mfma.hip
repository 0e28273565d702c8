// conv_mfma.hip — implicit-GEMM Conv3d 3x3x3 / stride 1 / pad 1 / dilation 1 on the fp32-input MFMA
// (v_mfma_f32_16x16x4_f32: exact fp32 fmaf chains at 157 TFLOP/s peak, the fp32 roof of gfx950), for the hot layers
// of the U-Net: forward, data-gradient (the same kernel on flipped/transposed weights) and weight-gradient.
//
//   forward   Y[v, co] = sum_{tap, ci} X[v + tap - 1, ci] * W[co, ci, tap]          GEMM  M = voxels, N = co, K = 27*ci
//   dgrad     dX[v, ci] = sum_{tap, co} dY[v + 1 - tap, co] * W[co, ci, tap]        same kernel, K = 27*co, N = ci
//   wgrad     dW[co, ci, tap] = sum_v X[v + tap - 1, ci] * dY[v, co]                GEMM  M = (tap, ci), N = co, K = voxels
//
// Data layout in HBM: NDHWC activations (voxel pitch ld), so the K = ci fibre of a voxel is contiguous.
// LDS: one workgroup stages the (4+2) x (8+2) x (16+2) input halo tile of a 16- (or 8-) channel chunk, [voxel][ci].
// MFMA mapping (forward): an M-tile is 16 consecutive voxels along W; lane l supplies A[i = l&15][k-group = l>>4].
//   The K order inside a 16-channel chunk is permuted so that k-group g owns channels 4g..4g+3: one ds_read_b128 per
//   lane feeds four consecutive MFMA k-steps (step s takes element s).  The packed weight image applies the same
//   permutation, and is laid out so that each wave reads its B fragments as one contiguous 1 KiB global load.
//   With 8-channel chunks two taps share a k-step (k-groups 0,1 = tap 2t, k-groups 2,3 = tap 2t+1).
// A wave owns one d-plane of the tile = 8 M-tiles x NT N-tiles of 16 output channels (8*NT accumulators of 4 VGPRs),
// so every B fragment is reused 8x and every A fragment NT x.
//
// Roofline: compute-bound for Cin*Cout >= 8*16 (AI 72..270 FLOP/B vs ridge ~20): the bound is the fp32
// MFMA peak; algorithmic FLOPs = 2 * N*D*H*W * Cin * Cout * 27 per pass.
#include "common.h"
#include "mfma_util.h"
#include <stdlib.h>
#include <type_traits>

namespace mri3d {

// conv_march.hip: forward / data gradient marching along d.  `force` = the explicit entry points (every geometry the kernel can
// compute); otherwise the dispatcher's own choice of the layers where it is the faster kernel.
bool conv_march_takes(const Mri3dConvGeom& g, bool dgrad, bool stats, bool force);
size_t conv_march_workspace_bytes(const Mri3dConvGeom& g, bool dgrad);
int conv_march_stat_blocks(const Mri3dConvGeom& g, bool force);
int conv_march_run(const Mri3dConvGeom& g, bool dgrad, bool force, const void* in_v, const float* w, const float* bias, void* out_v,
                   void* ws, size_t ws_bytes, hipStream_t s, double* stat_part, const void* second, int split, int second_ld);

constexpr int TD = 4, TH = 8, TW = 16;               // output tile (d, h, w)
constexpr int HD = TD + 2, HH = TH + 2, HW = TW + 2;  // halo tile
constexpr int HVOX = HD * HH * HW;                   // 1080 voxels

__host__ __device__ constexpr int tap_groups(int CK) { return CK == 16 ? 27 : 14; }

// Tap pairing of the two-taps-per-k-step kernels (8-channel fp32 chunks, 16-channel bf16 chunks): tap group tg carries taps
// pair_tap(tg, 0) in k-groups 0,1 and pair_tap(tg, 1) in k-groups 2,3 (27 = padding, zero weights).  The pairs are chosen so
// that the three groups of a CLASS differ only in kh, i.e. by ONE ROW of the halo tile:
//   groups 3c+kh (c = kd = 0..2):  (kd, kh, 0) | (kd, kh, 1)          groups 9+kh:  (0, kh, 2) | (1, kh, 2)
//   group 12:  (2, 0, 2) | (2, 1, 2)                                  group 13:     (2, 2, 2) | padding
// A-fragment (voxel rows) of group (class, kh), output row m == fragment of group (class, 0), row m + kh, so a wave needs 10
// LDS fragments per class instead of 3 x 8 (conv_mfma_fwd2_kernel, one N-tile): 56 instead of 112 ds_read_b128 per chunk.
__host__ __device__ constexpr int pair_tap(int tg, int half) {
    return tg < 9 ? ((tg / 3) * 3 + tg % 3) * 3 + half
                  : (tg < 12 ? (half * 3 + (tg - 9)) * 3 + 2 : (tg == 12 ? (6 + half) * 3 + 2 : (half == 0 ? 26 : 27)));
}

// ------------------------------------------------------------------ weight packing
// Wp[chunk][tg][nt][lane][s]:  value = W'(tap, kc, nc) with
//   nc = nt*16 + (lane & 15)
//   CK == 16: tap = tg,                 kc = chunk*16 + 4*(lane>>4) + s
//   CK ==  8: tap = pair_tap(tg, lane>>5), kc = chunk*8  + 4*((lane>>4)&1) + s   (tap 27 -> 0)
//   forward: W'(tap,kc,nc) = W[nc][kc][tap];  dgrad: W'(tap,kc,nc) = W[kc][nc][26 - tap]
__global__ void pack_w_mfma_kernel(const float* __restrict__ w, float* __restrict__ wp, int Co, int Ci, int dgrad,
                                   int CK, int NTT, int nchunks) {
    const int TG = tap_groups(CK);
    const int total = nchunks * TG * NTT * 256;
    const int Kc = dgrad ? Co : Ci, Nc = dgrad ? Ci : Co;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        int s = i & 3, lane = (i >> 2) & 63;
        int t = i >> 8;
        int nt = t % NTT;
        t /= NTT;
        int tg = t % TG;
        int chunk = t / TG;
        int nc = nt * 16 + (lane & 15);
        int tap, kc;
        if (CK == 16) {
            tap = tg;
            kc = chunk * 16 + 4 * (lane >> 4) + s;
        } else {
            tap = pair_tap(tg, lane >> 5);
            kc = chunk * 8 + 4 * ((lane >> 4) & 1) + s;
        }
        float v = 0.f;
        if (tap < 27 && nc < Nc && kc < Kc) {
            v = dgrad ? w[((size_t)kc * Ci + nc) * 27 + (26 - tap)] : w[((size_t)nc * Ci + kc) * 27 + tap];
        }
        wp[i] = v;
    }
}

// bf16 image for the 16x16x32 MFMA (conv_mfma_fwd2_kernel<bf16_t>): a 32-byte voxel slice holds 16 channels, so a
// chunk is 16 channels and a lane's 16-byte fragment is 8 consecutive channels of one tap:
//   Wp[chunk][tg][nt][lane][s], s = 0..7:  tap = pair_tap(tg, lane>>5), kc = chunk*16 + 8*((lane>>4)&1) + s  (tap 27 -> 0)
__global__ void pack_w_mfma_bf16_kernel(const float* __restrict__ w, bf16_t* __restrict__ wp, int Co, int Ci, int dgrad,
                                        int NTT, int nchunks) {
    constexpr int TG = 14;
    const int total = nchunks * TG * NTT * 512;
    const int Kc = dgrad ? Co : Ci, Nc = dgrad ? Ci : Co;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int s = i & 7, lane = (i >> 3) & 63;
        int t = i >> 9;
        const int nt = t % NTT;
        t /= NTT;
        const int tg = t % TG, chunk = t / TG;
        const int nc = nt * 16 + (lane & 15);
        const int tap = pair_tap(tg, lane >> 5);
        const int kc = chunk * 16 + 8 * ((lane >> 4) & 1) + s;
        float v = 0.f;
        if (tap < 27 && nc < Nc && kc < Kc)
            v = dgrad ? w[((size_t)kc * Ci + nc) * 27 + (26 - tap)] : w[((size_t)nc * Ci + kc) * 27 + tap];
        wp[i] = (bf16_t)v;
    }
}

// Weight image of the 8-output-channel variant (conv_mfma_fwd2_kernel<.., N8>): Wp[chunk][group 0..9][lane][s].  Row li of the
// operand = channel li & 7 of row half hs = li >> 3; k-slot ks = lane >> 5 as in the images above.
//   groups 2u, 2u+1 (class u = 0..3: k-slot taps (kd, kw) = (u, ks) for u < 3, (ks, 2) for u = 3):
//       2u   (PAIR)   kh = hs         2u+1 (SINGLE) kh = 2, row half 0 only
//   group 8: taps (kd 2, kh ks, kw 2), group 9: tap 26 in k-slot 0 — row half 0 only (the old groups 12 and 13)
template <typename WT>
__global__ void pack_w_mfma_n8_kernel(const float* __restrict__ w, WT* __restrict__ wp, int Co, int Ci, int dgrad, int nchunks) {
    constexpr int PE = 16 / sizeof(WT);   // elements per lane fragment (4 fp32 / 8 bf16)
    const int total = nchunks * 10 * 64 * PE;
    const int Kc = dgrad ? Co : Ci;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int sidx = i % PE, lane = (i / PE) & 63;
        int t = i / (PE * 64);
        const int grp = t % 10, chunk = t / 10;
        const int li = lane & 15, nc = li & 7, hs = li >> 3, ks = lane >> 5;
        const int kc = chunk * 2 * PE + PE * ((lane >> 4) & 1) + sidx;
        int tap = -1;
        if (grp < 8) {
            const int u = grp >> 1, single = grp & 1;
            const int kd = u < 3 ? u : ks, kw = u < 3 ? ks : 2;
            const int kh = single ? 2 : hs;
            if (!(single && hs)) tap = (kd * 3 + kh) * 3 + kw;
        } else if (grp == 8) {
            if (!hs) tap = (2 * 3 + ks) * 3 + 2;
        } else {
            if (!hs && ks == 0) tap = 26;
        }
        float v = 0.f;
        if (tap >= 0 && kc < Kc) v = dgrad ? w[((size_t)kc * Ci + nc) * 27 + (26 - tap)] : w[((size_t)nc * Ci + kc) * 27 + tap];
        wp[i] = (WT)v;
    }
}

#if defined(MRI3D_EXPERIMENT_STAMPS)   // tuning builds: in-kernel phase stamps of wave 0 of workgroup 0 (cdna_hip_programming.md §7)
__device__ unsigned long long g_stamps[8];
__device__ unsigned long long g_block_span[2 * 1024];   // per workgroup of the LAST launch: 100 MHz ticks at loop entry and exit
extern "C" void mri3d_debug_stamps(unsigned long long* out, int reset) {
    if (reset) {
        unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z, sizeof(z));
    } else {
        (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 8);
    }
}
extern "C" void mri3d_debug_block_spans(unsigned long long* out) {
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_block_span), sizeof(unsigned long long) * 2 * 1024);
}
// the phase sums stay in (scalar) registers until the kernel's end: a stamp must not add memory operations or waits to the loop
// (a stamp is pinned by scheduling barriers: s_memtime depends on nothing, and hipcc otherwise moves it across the MFMAs it brackets)
#define MRI3D_STAMP(var)                     \
    __builtin_amdgcn_sched_barrier(0);       \
    unsigned long long var = __builtin_readcyclecounter(); \
    __builtin_amdgcn_sched_barrier(0)
#define MRI3D_STAMP_ADD(slot, a, b) (stamp_acc[slot] += (b) - (a))
#else
#define MRI3D_STAMP(var)
#define MRI3D_STAMP_ADD(slot, a, b)
#endif

// ------------------------------------------------------------------ forward / dgrad kernel
// Persistent workgroups + double-buffered LDS + staging folded into the tap loop.
//   A first version (round 1, removed) alternated "stage a chunk" and "27 tap groups of MFMA" with a barrier pair in between; the two
//   workgroups resident on a CU ran in lock-step, so the staging time (1.0 of 4.1 ms on the 48->16 layer, measured by
//   ablation) was NOT hidden.  This kernel makes each workgroup self-overlapping:
//     * K is consumed in 8-channel chunks (two taps share an MFMA k-step: 14 tap groups per chunk), so TWO halo tiles
//       fit in LDS (2 x 34.5 KB) with two workgroups per CU;
//     * while chunk i is multiplied out of buffer i&1, every lane also fetches its 9 16-byte pieces of chunk i+1 (of the
//       same tile or the next one of this workgroup's range) — one global load per tap group, written to buffer
//       (i+1)&1 three tap groups later — so global latency, LDS writes and MFMAs overlap; one barrier per chunk;
//     * operands are passed to the MFMA as (weights, voxels): the accumulator then holds 4 consecutive output channels
//       of one voxel per lane and the epilogue is a fully coalesced 16-byte store per lane (1 KiB per wave).
constexpr int kStg = (HVOX * 2 + 255) / 256;  // 16-byte staging pieces per lane per 8-channel chunk (9)

#ifndef MRI3D_DMA_PPT
#define MRI3D_DMA_PPT 1   // DMA pieces issued per tap group (9 pieces per chunk)
#endif

// T = float: 8-channel chunks, four 16x16x4 fp32 MFMAs per tap group.  T = bf16_t: the SAME byte geometry (a 32-byte voxel
// slice = 16 channels, 16-byte pieces, identical staging and LDS addressing) with one v_mfma_f32_16x16x32_bf16 per tap
// group; the kernel is then bound by the LDS operand reads (the bf16 3x3x3 layers are memory-bound).
// `wp` is the packed weight image (fp32 or bf16), addressed in 16-byte fragments.
// STATS (forward only): the epilogue also accumulates, per output channel, sum(a) and sum(a^2) of the convolution result a
// WITHOUT its bias over the voxels of the volume — the BatchNorm batch statistics of y = a + bias (shift = bias), so that the
// statistics pass over y (one full read of every conv output, 0.5 ms per step of the U-Net) is not needed.  fp32 over a wave's
// 8 x 16 voxels of a tile, float64 from there on: per wave in LDS, one partial per workgroup in `stat_part`
// [gridDim.x][Nc][2], summed in a fixed order by norm_stats_finalize_kernel (deterministic: the tile -> workgroup map is static).
// N8 (exactly 8 output channels: the first level of Modified3DUNet, modified_3dunet.py:33-55, and the 8 -> 16 data gradient of
// the U-Net): a 16-row weight operand would be half zeros.  Its rows 8..15 take the SAME channels for the tap one halo row further
// (kh + 1) instead: with the voxel fragment of halo row i, rows 0..7 add to output row i (tap kh) and rows 8..15 to output row
// i - 1 (tap kh + 1).  A class of three tap groups (kh = 0, 1, 2: 3 x 8 row-MFMAs) becomes a PAIR group over halo rows 0..8 and
// a SINGLE group (kh = 2) over rows 2..9: 17 row-MFMAs; 84 instead of 112 per chunk.  Nine accumulators (halo rows 0..8); the
// epilogue adds the upper half of accumulator r + 1 (lanes 32..63) to the lower half of accumulator r (lanes 0..31).
// Packed weights: pack_w_mfma_n8_kernel, 10 groups per chunk.
template <typename T, int NT, bool STATS, bool N8 = false>
__global__ void __launch_bounds__(256, 2)
conv_mfma_fwd2_kernel(const T* __restrict__ x, const float* __restrict__ wp, const float* __restrict__ bias,
                      T* __restrict__ y, int N, int D, int H, int W, int Kc, int x_ld, int Nc, int y_ld, int NTT,
                      int gy, int tilesD, int tilesH, int tilesW, int ntiles, double* __restrict__ stat_part,
                      const T* __restrict__ x2, int x2_ld, int ksplit, T* __restrict__ y2, int y2_ld, int nsplit) {
    // x2 (forward of a conv over cat((x, x2), channels)): input channels >= ksplit live in the second tensor with its own pitch —
    // a chunk's DMA resource is simply based in the tensor that holds it, so torch.cat never materialises (unet.UNet decoder,
    // segmentation/routine.py:346-356).  y2 (its data gradient): output channels >= nsplit are written to the second tensor, so
    // both gradients come out dense.  ksplit, nsplit are multiples of 16; x2 = y2 = nullptr: one tensor each.
    constexpr bool kBf16 = sizeof(T) == 2;
    constexpr int CK = 32 / sizeof(T);   // channels per 32-byte chunk (8 fp32 / 16 bf16)
    constexpr int PE = 16 / sizeof(T);   // channels per 16-byte piece
    constexpr int CP = 8, TG = N8 ? 10 : 14;   // CP: LDS voxel pitch in floats (32 bytes); TG: weight fragments (groups) per chunk
    static_assert(!N8 || (NT == 1 && !STATS), "the 8-channel variant has one N-tile and no fused statistics");
    constexpr int AR = TH + (N8 ? 1 : 0);      // accumulator rows
    constexpr int BUF = kStg * 256 * 4;  // floats per LDS buffer: the halo tile rounded up to kStg 16-byte pieces per lane
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform BY CONSTRUCTION: told to hipcc, so that everything
                                                               // derived from it (output plane, tile base pointers) stays scalar
    const int li = lane & 15, kq = lane >> 4;
    const int nchunks = (Kc + CK - 1) / CK;   // bf16 with Kc % 16 == 8: the last chunk's upper piece is zero-filled
    // Tile -> workgroup map (speed only): blocks b and b+8 share an XCD (round-robin dispatch), so XCD k = b & 7 owns the
    // contiguous tile range [ntiles*k/8, ntiles*(k+1)/8) and its workgroups take those tiles round-robin: at any moment
    // the ~64 workgroups of an XCD work on ~64 CONSECUTIVE tiles, whose shared halo voxels then hit that XCD's L2
    // instead of being re-fetched over the fabric.
    const int NX = gridDim.x < 8 ? (int)gridDim.x : 8;                  // XCD groups that actually received a block
    const int xcd = blockIdx.x % NX, wslot = blockIdx.x / NX;
    const int wper = ((int)gridDim.x - xcd + NX - 1) / NX;             // workgroups living on this XCD
    const int r_lo = (int)(((int64_t)ntiles * xcd) / NX), r_hi = (int)(((int64_t)ntiles * (xcd + 1)) / NX);
    const int my_tiles = (r_hi - r_lo - wslot + wper - 1) / wper;      // tiles r_lo + wslot + k * wper < r_hi
    const int nitems = (r_lo + wslot < r_hi ? my_tiles : 0) * nchunks;
    double* const stat_lds = reinterpret_cast<double*>(lds + 2 * BUF);   // [4 waves][NTT * 16 channels][2]
    if constexpr (STATS) {
        if (nitems <= 0) {   // a workgroup without tiles still owns a partial: zeros
            for (int i = tid; i < Nc * 2; i += 256) stat_part[(size_t)blockIdx.x * Nc * 2 + i] = 0.0;
            return;
        }
        for (int i = tid; i < 4 * NTT * 16 * 2; i += 256) stat_lds[i] = 0.0;   // visible after the prologue's barrier
    }
    if (nitems <= 0) return;

    // per-lane staging geometry (independent of the item): piece j covers halo voxel (j*256+tid)>>1, channel quad &1
    // (pieces past the tile's end alias voxel 0 and land in the buffer's padding: no lane ever branches)
    int srel[kStg];  // packed (dz, hy, wx)
#pragma unroll
    for (int j = 0; j < kStg; ++j) {
        const int idx = j * 256 + tid;
        const int v = idx < HVOX * 2 ? idx >> 1 : 0;
        const int wx = v % HW, t2 = v / HW;
        srel[j] = ((t2 / HH) << 16) | ((t2 % HH) << 8) | wx;
    }
    // ... and its BYTE offset from the item's halo origin (voxel (d0-1, h0-1, w0-1), channel ch*CK): the DMA's buffer resource
    // is based at that origin, so an in-volume piece's offset is this per-lane constant — no per-piece integer multiplies,
    // clamps or 64-bit arithmetic — and an out-of-volume piece gets the out-of-range offset that makes the DMA write zeros.
    unsigned vrel[kStg];   // voxel index relative to the halo origin; byte offset = vrel * pitch bytes + the lane's piece (one v_mad)
#pragma unroll
    for (int j = 0; j < kStg; ++j) {
        const int r = srel[j];
        vrel[j] = (unsigned)(((r >> 16) * H + ((r >> 8) & 0xff)) * W + (r & 0xff));
    }
    const unsigned pieceb = (unsigned)(PE * (tid & 1)) * (unsigned)sizeof(T);

    struct Item { int n, d0, h0, w0, nt0, ch; };
    auto decode = [&](int it) -> Item {
        Item r;
        int tile = r_lo + wslot + (it / nchunks) * wper;
        r.ch = it % nchunks;
        r.nt0 = (tile % gy) * NT;
        tile /= gy;
        r.w0 = (tile % tilesW) * TW;
        tile /= tilesW;
        // tiles run w-fastest, then d, then h: the d-neighbours (which share 2 of 6 halo planes) are tilesW apart and run
        // side by side on one XCD; measured against (w, h, d): bf16 forward 48->16 0.658 -> 0.621 ms (profiles/r02_tile_order.txt)
        r.d0 = (tile % tilesD) * TD;
        tile /= tilesD;
        r.h0 = (tile % tilesH) * TH;
        r.n = tile / tilesH;

        return r;
    };
    // A workgroup's tiles are `wper` apart: the next tile's coordinates come from adding wper's mixed-radix digits (N-block, w, d,
    // h, n) with carries — ~15 scalar instructions instead of decode()'s five run-time divisions (~150).  Measured by ablation
    // (decode replaced by a constant): the divisions cost 5 % of the fp32 16 -> 16 forward, 12 % of 8 -> 16 and 22 % of the bf16
    // 16 -> 16 forward, whose tiles are a single chunk (one decode per item).
    int sdig[5];
    {
        int q = wper;
        sdig[0] = q % gy;
        q /= gy;
        sdig[1] = q % tilesW;
        q /= tilesW;
        sdig[2] = q % tilesD;
        q /= tilesD;
        sdig[3] = q % tilesH;
        sdig[4] = q / tilesH;
    }
    auto advance = [&](const Item& c) -> Item {   // chunk 0 of the tile `wper` after c's
        Item r;
        r.ch = 0;
        int a = c.nt0 / NT + sdig[0];
        int cy = a >= gy;
        r.nt0 = (a - (cy ? gy : 0)) * NT;
        a = c.w0 / TW + sdig[1] + cy;
        cy = a >= tilesW;
        r.w0 = (a - (cy ? tilesW : 0)) * TW;
        a = c.d0 / TD + sdig[2] + cy;
        cy = a >= tilesD;
        r.d0 = (a - (cy ? tilesD : 0)) * TD;
        a = c.h0 / TH + sdig[3] + cy;
        cy = a >= tilesH;
        r.h0 = (a - (cy ? tilesH : 0)) * TH;
        r.n = c.n + sdig[4] + cy;
        return r;
    };
    const unsigned lds_wave = __builtin_amdgcn_readfirstlane((unsigned)(size_t)lds + (unsigned)wv * 1024u);
    // Stage one item (a 32-byte channel chunk of a halo tile) into LDS buffer `bsel`: kStg DMA pieces per lane, piece j of lane
    // `tid` = 16-byte piece j*256 + tid of the [halo voxel][32 B] image — lane-linear, which is what an LDS-DMA writes.
    // Tiles whose halo lies inside the volume (70 % at 160x192x160) take the wave-uniform fast path: the offsets are the
    // per-lane constants, no VALU work at all.  (Pieces past the tile's end alias voxel 0 and land in the buffer's padding.)
    struct Stage { i32x4 rs; unsigned dst, ldb; bool interior, on; };
    auto stage_open = [&](const Item& it, int bsel, bool on) -> Stage {
        Stage st;
        const bool second = x2 != nullptr && it.ch * CK >= ksplit;   // wave-uniform: which tensor holds this chunk
        const T* xs = second ? x2 : x;
        const int ld = second ? x2_ld : x_ld, c0 = second ? it.ch * CK - ksplit : it.ch * CK;
        st.ldb = (unsigned)ld * (unsigned)sizeof(T);
        const unsigned long long org =
            (unsigned long long)(xs + (((((int64_t)it.n * D + it.d0 - 1) * H + it.h0 - 1) * W + it.w0 - 1) * ld + c0));
        // raw buffer resource: base (48 bits), stride 0, num_records, gfx9 raw-dword format
        st.rs[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)(org & 0xffffffffu));
        st.rs[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)((org >> 32) & 0xffffu));
        st.rs[2] = (int)kDmaRecords;
        st.rs[3] = 0x00020000;
        st.dst = lds_wave + (unsigned)bsel * (unsigned)(BUF * 4);
        st.interior = it.d0 >= 1 && it.d0 + TD < D && it.h0 >= 1 && it.h0 + TH < H && it.w0 >= 1 && it.w0 + TW < W &&
                      (it.ch + 1) * CK <= Kc;
        st.on = on;
        return st;
    };
    auto stage_piece = [&](const Item& it, const Stage& st, int j) {   // wave-uniform branches only
        if (!st.on) return;
        if (st.interior) {
            lds_dma16(__umul24(vrel[j], st.ldb) + pieceb, st.rs, st.dst + j * 4096);   // v_mad_u32_u24: both factors < 2^24
        } else {
            const int r = srel[j];
            const int gd = it.d0 - 1 + (r >> 16), gh = it.h0 - 1 + ((r >> 8) & 0xff), gw = it.w0 - 1 + (r & 0xff);
            const bool ok = (unsigned)gd < (unsigned)D && (unsigned)gh < (unsigned)H && (unsigned)gw < (unsigned)W &&
                            it.ch * CK + PE * (tid & 1) < Kc;
            lds_dma16(ok ? __umul24(vrel[j], st.ldb) + pieceb : kDmaOob, st.rs, st.dst + j * 4096);
        }
    };
    // pieces issued in front of tap group tg: kPpt per group, so that a wave's issue slots (~60-180 cycles per piece) are
    // spread behind MFMAs instead of delaying the chunk's first fragments
    constexpr int kPpt = MRI3D_DMA_PPT;
    auto stage_group = [&](const Item& it, const Stage& st, int tg) {
#pragma unroll
        for (int j = tg * kPpt; j < (tg + 1) * kPpt && j < kStg; ++j) stage_piece(it, st, j);
    };
    auto weights_of = [&](const Item& it) -> const float* {
        return wp + ((size_t)it.ch * TG * NTT + it.nt0) * 256 + lane * 4;
    };
    const size_t wstep = (size_t)NTT * 256;

#if defined(MRI3D_EXPERIMENT_STAMPS)   // the clock the chip holds in this kernel: shader cycles / 100 MHz reference ticks
    const unsigned long long clk0 = __builtin_amdgcn_s_memtime(), ref0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    // prologue: item 0 -> buffer 0 (and, one N-tile: all 14 weight fragments of its chunk)
    Item cur = decode(0);
    f32x4 bqa[NT == 1 ? TG : 1];   // NT = 1: the weight fragments of the WHOLE chunk; fragment tg is re-loaded for the next item
                                   // right after tap group tg has used it — a full chunk of latency cover, and no weight load
                                   // ever queues behind the DMA pieces it does not depend on (VMEM returns in order)
    if constexpr (NT == 1) {
        const float* w0p = weights_of(cur);
#pragma unroll
        for (int tg = 0; tg < TG; ++tg) bqa[tg] = *reinterpret_cast<const f32x4*>(w0p + (size_t)tg * wstep);
    }
    {
        const Stage st0 = stage_open(cur, 0, true);
#pragma unroll
        for (int j = 0; j < kStg; ++j) stage_piece(cur, st0, j);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    f32x4 acc[AR][NT];
#pragma unroll
    for (int m = 0; m < AR; ++m)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[m][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    float4 bias0[NT];   // the lane's bias quads when the kernel has one N-block (gy == 1: nt0 is always 0)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        bias0[nt] = make_float4(0.f, 0.f, 0.f, 0.f);
        const int co = nt * 16 + 4 * kq;
        if (NT == 1 && bias && gy == 1 && co < Nc) {   // (two N-tiles: no registers to spare, the bias is loaded per tile)
            bias0[nt].x = bias[co];
            if (co + 1 < Nc) bias0[nt].y = bias[co + 1];
            if (co + 2 < Nc) bias0[nt].z = bias[co + 2];
            if (co + 3 < Nc) bias0[nt].w = bias[co + 3];
        }
    }
    // A-fragment LDS offsets of a tap group (k-groups 0,1 = pair_tap(tg, 0); k-groups 2,3 = pair_tap(tg, 1), tap 27 = zero weights)
    auto a_off = [&](int tg) -> int {
        const int ta = pair_tap(tg, 0), tb = pair_tap(tg, 1) < 27 ? pair_tap(tg, 1) : 26;
        const int oa = ((ta / 9) * HH + (ta / 3) % 3) * HW + ta % 3;
        const int ob = ((tb / 9) * HH + (tb / 3) % 3) * HW + tb % 3;
        return ((wv * HH) * HW + li + ((kq >> 1) ? ob : oa)) * CP + 4 * (kq & 1);
    };

    for (int it = 0; it < nitems; ++it) {
        MRI3D_STAMP(t_item);
        const float* bufc = lds + (it & 1) * BUF;
        const bool has_next = it + 1 < nitems;
        Item nxt = cur;
        // the next item is the next chunk of the same tile, or chunk 0 of this workgroup's next tile (advance(): digit adds)
        if (has_next) {
            if (cur.ch + 1 < nchunks) nxt.ch = cur.ch + 1;
            else nxt = advance(cur);
        }
        if constexpr (N8) {
            const Stage stn = stage_open(nxt, (it + 1) & 1, has_next);
            const float* wtn = weights_of(nxt);
            // units: four classes (a PAIR and a SINGLE group on one set of 10 row fragments), then the old groups 12 and 13
            constexpr int NU = 6;
            f32x4 fr[2][TH + 2];
            {
                const int o0 = a_off(0);
#pragma unroll
                for (int i = 0; i < TH + 2; ++i) fr[0][i] = *reinterpret_cast<const f32x4*>(bufc + o0 + i * HW * CP);
            }
#pragma unroll
            for (int g = 0; g < TG; ++g) {
                const int u = g < 8 ? g / 2 : g - 4, ng = u < 4 ? 2 : 1, gi = u < 4 ? g - 2 * u : 0;
                stage_group(nxt, stn, g);
                if (g >= 1) bqa[g - 1] = *reinterpret_cast<const f32x4*>(wtn + (size_t)(g - 1) * wstep);   // next item's
                if (u + 1 < NU) {   // this group's share of the next unit's fragments
                    const int nu = u + 1, nfirst = nu < 4 ? 3 * nu : nu + 8, nf = nu < 4 ? TH + 2 : TH;
                    const int on = a_off(nfirst);
#pragma unroll
                    for (int i = 0; i < TH + 2; ++i)
                        if (i >= gi * nf / ng && i < (gi + 1) * nf / ng)
                            fr[nu & 1][i] = *reinterpret_cast<const f32x4*>(bufc + on + i * HW * CP);
                }
                __builtin_amdgcn_sched_barrier(0);  // keep the prefetches above this group's MFMAs
                // PAIR group: halo rows 0..8 -> accumulators 0..8; SINGLE group (kh = 2): halo rows 2..9 -> accumulators 0..7;
                // groups 8, 9: halo rows 0..7 -> accumulators 0..7
                const int r0 = (u < 4 && gi == 1) ? 2 : 0, nr = (u < 4 && gi == 0) ? TH + 1 : TH;
                if constexpr (kBf16) {
#pragma unroll
                    for (int m = 0; m < TH + 1; ++m)
                        if (m < nr)
                            acc[m][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, bqa[g]),
                                                                                __builtin_bit_cast(bf16x8_t, fr[u & 1][m + r0]),
                                                                                acc[m][0], 0, 0, 0);
                } else {
#pragma unroll
                    for (int m = 0; m < TH; m += 2)
#pragma unroll
                        for (int s4 = 0; s4 < 4; ++s4) {
                            acc[m][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(bqa[g][s4], fr[u & 1][m + r0][s4], acc[m][0], 0, 0, 0);
                            acc[m + 1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(bqa[g][s4], fr[u & 1][m + 1 + r0][s4], acc[m + 1][0], 0, 0, 0);
                        }
                    if (nr == TH + 1) {
#pragma unroll
                        for (int s4 = 0; s4 < 4; ++s4)
                            acc[TH][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(bqa[g][s4], fr[u & 1][TH][s4], acc[TH][0], 0, 0, 0);
                    }
                }
            }
            bqa[TG - 1] = *reinterpret_cast<const f32x4*>(wtn + (size_t)(TG - 1) * wstep);
            if (cur.ch == nchunks - 1) {
                // fold: output row r = lower half of accumulator r + upper half (lanes 32..63 -> 0..31) of accumulator r + 1
#pragma unroll
                for (int m = 0; m < TH; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[m][0][r] += __shfl_down(acc[m + 1][0][r], 32, 64);
            }
        } else if constexpr (NT == 1) {
            // One N-tile: every A-fragment feeds a single MFMA chain, so the LDS reads are the largest non-MFMA cost.  The tap
            // groups come in UNITS: four classes of three groups that differ only by one halo row (pair_tap), then groups 12
            // and 13.  A class needs 10 row fragments (rows 0..9 of its kh = 0 group; group kh, output row m uses row m + kh)
            // instead of 3 x 8; the next unit's fragments are fetched while this unit is multiplied (two register sets).
            const Stage stn = stage_open(nxt, (it + 1) & 1, has_next);   // the next chunk: 9 DMA pieces per lane, landed by the barrier
            const float* wtn = weights_of(nxt);             // (the last item re-reads its own: the loads stay unconditional)
            constexpr int NU = 6;
            f32x4 fr[2][TH + 2];
            {
                const int o0 = a_off(0);
#pragma unroll
                for (int i = 0; i < TH + 2; ++i) fr[0][i] = *reinterpret_cast<const f32x4*>(bufc + o0 + i * HW * CP);
            }
#pragma unroll
            for (int tg = 0; tg < TG; ++tg) {
                const int u = tg < 12 ? tg / 3 : tg - 8, ufirst = u < 4 ? 3 * u : u + 8, ng = u < 4 ? 3 : 1;
                const int kh = tg - ufirst;   // row shift inside the class (0 for the single-group units)
                stage_group(nxt, stn, tg);
                if (tg >= 1) bqa[tg - 1] = *reinterpret_cast<const f32x4*>(wtn + (size_t)(tg - 1) * wstep);   // next item's
                if (u + 1 < NU) {   // this group's share of the next unit's fragments
                    const int nu = u + 1, nfirst = nu < 4 ? 3 * nu : nu + 8, nf = nu < 4 ? TH + 2 : TH;
                    const int on = a_off(nfirst);
#pragma unroll
                    for (int i = 0; i < TH + 2; ++i)
                        if (i >= kh * nf / ng && i < (kh + 1) * nf / ng)
                            fr[nu & 1][i] = *reinterpret_cast<const f32x4*>(bufc + on + i * HW * CP);
                }
                __builtin_amdgcn_sched_barrier(0);  // keep the prefetches above this tap group's MFMAs
                if constexpr (kBf16) {
#pragma unroll
                    for (int m = 0; m < TH; ++m)
                        acc[m][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, bqa[tg]),
                                                                            __builtin_bit_cast(bf16x8_t, fr[u & 1][m + kh]),
                                                                            acc[m][0], 0, 0, 0);
                } else {
                    // accumulator reuse distance 2 (two rows alternate over the four k-steps): the fp32 MFMA sustains its peak
                    // when an accumulator comes back after <= 3 or >= 16 instructions, not after 4..8 (tools/microbench)
#pragma unroll
                    for (int m = 0; m < TH; m += 2)
#pragma unroll
                        for (int s = 0; s < 4; ++s) {
                            acc[m][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(bqa[tg][s], fr[u & 1][m + kh][s], acc[m][0], 0, 0, 0);
                            acc[m + 1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(bqa[tg][s], fr[u & 1][m + 1 + kh][s], acc[m + 1][0], 0, 0, 0);
                        }
                }
            }
            bqa[TG - 1] = *reinterpret_cast<const f32x4*>(wtn + (size_t)(TG - 1) * wstep);
        } else {
        // Two N-tiles: a 3-deep ring of weight fragments (two tap groups of lead).  The ring is primed BEFORE the DMA pieces are
        // issued, so that the first tap group does not wait for them; the loads of tap groups 2.. queue behind the pieces,
        // which have two tap groups (fp32: 3 us) to land.
        const float* wt = weights_of(cur);
        f32x4 bq[3][NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            bq[0][nt] = *reinterpret_cast<const f32x4*>(wt + nt * 256);
            bq[1][nt] = *reinterpret_cast<const f32x4*>(wt + wstep + nt * 256);
        }
        const Stage stn = stage_open(nxt, (it + 1) & 1, has_next);
        f32x4 aq[2][TH];
        {
            const int o0 = a_off(0);
#pragma unroll
            for (int m = 0; m < TH; ++m) aq[0][m] = *reinterpret_cast<const f32x4*>(bufc + o0 + m * HW * CP);
        }
#pragma unroll
        for (int tg = 0; tg < TG; ++tg) {
            const int cb = tg % 3, nb = (tg + 2) % 3, ac = tg & 1, an = (tg + 1) & 1;
            stage_group(nxt, stn, tg);
            if (tg + 2 < TG) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    bq[nb][nt] = *reinterpret_cast<const f32x4*>(wt + (size_t)(tg + 2) * wstep + nt * 256);
            }
            if (tg + 1 < TG) {
                const int o1 = a_off(tg + 1);
#pragma unroll
                for (int m = 0; m < TH; ++m) aq[an][m] = *reinterpret_cast<const f32x4*>(bufc + o1 + m * HW * CP);
            }
            __builtin_amdgcn_sched_barrier(0);  // keep the prefetches above this tap group's MFMAs
            if constexpr (kBf16) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int m = 0; m < TH; ++m)
                        acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, bq[cb][nt]),
                                                                             __builtin_bit_cast(bf16x8_t, aq[ac][m]),
                                                                             acc[m][nt], 0, 0, 0);
            } else {
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int m = 0; m < TH; ++m)
                            acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bq[cb][nt][s], aq[ac][m][s], acc[m][nt], 0, 0, 0);
            }
        }
        }

        MRI3D_STAMP(t_mfma);
        MRI3D_STAMP_ADD(0, t_item, t_mfma);
        if (cur.ch == nchunks - 1) {
            // epilogue: lane holds channels 4*kq..4*kq+3 of voxel li of every 16x16 tile.  Everything that does not depend on the
            // lane is kept scalar — the output plane (wv is wave-uniform), which tensor an N-tile goes to, the tile's base pointer, the
            // row-in-volume tests — and the lane's own test (its channels and its voxel column exist) is made ONCE around the eight
            // rows: a one-chunk tile (every item of the 16-channel bf16 layers) pays this block per item.
            const int od = cur.d0 + wv;
            if (od < D) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int cob = (cur.nt0 + nt) * 16;   // wave-uniform
                    if (cob >= Nc) continue;
                    const bool second = y2 != nullptr && cob >= nsplit;   // nsplit % 16 == 0: an N-tile lives in one tensor
                    T* const yd = second ? y2 : y;
                    const int yld = second ? y2_ld : y_ld, cbase = second ? cob - nsplit : cob;
                    const int co = cob + 4 * kq;
                    const bool vec = (co + 3 < Nc) && ((yld & 3) == 0);
                    float4 bv = bias0[nt];   // one N-tile and one N-block: loaded once per kernel
                    if (bias && (NT > 1 || gy > 1) && co < Nc) {
                        bv = make_float4(0.f, 0.f, 0.f, 0.f);
                        bv.x = bias[co];
                        if (co + 1 < Nc) bv.y = bias[co + 1];
                        if (co + 2 < Nc) bv.z = bias[co + 2];
                        if (co + 3 < Nc) bv.w = bias[co + 3];
                    }
                    // consume the loads HERE on every path: a bias register still "pending" at the loop's back edge makes hipcc
                    // wait vmcnt(0) at its next reuse — at the top of the next item, right behind the freshly issued DMA pieces
                    if (bias && (NT > 1 || gy > 1)) asm volatile("" ::"v"(bv.x), "v"(bv.y), "v"(bv.z), "v"(bv.w));
                    // scalar 64-bit tile base + 32-bit lane / row offsets
                    T* const ytile = yd + (((((int64_t)cur.n * D + od) * H + cur.h0) * W + cur.w0) * yld + cbase);
                    const unsigned lane_off = (unsigned)(li * yld + 4 * kq), row_step = (unsigned)(W * yld);
                    if (co < Nc && cur.w0 + li < W) {
                        if ((Nc & 3) == 0 && (yld & 3) == 0) {   // wave-uniform: one vector store per row, nothing else
#pragma unroll
                            for (int m = 0; m < TH; ++m) {
                                if (cur.h0 + m < H) {   // wave-uniform
                                    const f32x4 a = acc[m][nt];
                                    stf4(ytile + (lane_off + (unsigned)m * row_step), make_float4(a[0] + bv.x, a[1] + bv.y, a[2] + bv.z, a[3] + bv.w));
                                }
                            }
                        } else {
#pragma unroll
                            for (int m = 0; m < TH; ++m) {
                                if (cur.h0 + m < H) {
                                    T* yp = ytile + (lane_off + (unsigned)m * row_step);
                                    const f32x4 a = acc[m][nt];
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
                    }
                }
            }
            if constexpr (STATS) {
                const bool vok = cur.d0 + wv < D && cur.w0 + li < W;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int m = 0; m < TH; ++m) {
                        const bool ok = vok && cur.h0 + m < H;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float a = ok ? acc[m][nt][r] : 0.f;
                            s1[r] += a;
                            s2[r] += a * a;
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        s1[r] = row_sum16(s1[r]);
                        s2[r] = row_sum16(s2[r]);
                    }
                    if (li == 0) {   // one lane per k-group: channels 4*kq .. 4*kq+3 of this N-tile, this wave's own LDS slot
                        double* slot = stat_lds + ((size_t)wv * NTT * 16 + (cur.nt0 + nt) * 16 + 4 * kq) * 2;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            slot[2 * r] += (double)s1[r];
                            slot[2 * r + 1] += (double)s2[r];
                        }
                    }
                }
            }
#pragma unroll
            for (int m = 0; m < AR; ++m)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[m][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        // The DMA pieces of the next item are OLDER than the >= 12 weight loads issued after them in this iteration, and VMEM
        // returns in order: at most that many operations outstanding means every piece has landed in LDS.
        MRI3D_STAMP(t_epi);
        MRI3D_STAMP_ADD(1, t_mfma, t_epi);
        constexpr int kYounger = N8 ? 2 : 6;   // VMEM operations certainly issued after the last DMA piece (weight reloads)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kYounger) : "memory");
        MRI3D_STAMP(t_wait);
        MRI3D_STAMP_ADD(2, t_epi, t_wait);
        __syncthreads();  // buffer (it+1)&1 is complete; buffer it&1 may be overwritten from the next iteration on
        MRI3D_STAMP(t_bar);
        MRI3D_STAMP_ADD(3, t_wait, t_bar);
        MRI3D_STAMP_ADD(4, t_item, t_bar);
        MRI3D_STAMP_ADD(5, 0ull, 1ull);
        cur = nxt;
    }
#if defined(MRI3D_EXPERIMENT_STAMPS)
    {
        const unsigned long long clk1 = __builtin_amdgcn_s_memtime(), ref1 = __builtin_amdgcn_s_memrealtime();
        MRI3D_STAMP_ADD(6, clk0, clk1);
        MRI3D_STAMP_ADD(7, ref0, ref1);
        if (blockIdx.x == 0 && threadIdx.x == 0)
            for (int i = 0; i < 8; ++i) g_stamps[i] += stamp_acc[i];
        if (threadIdx.x == 0 && blockIdx.x < 1024) {
            g_block_span[2 * blockIdx.x] = ref0;
            g_block_span[2 * blockIdx.x + 1] = ref1;
        }
    }
#endif
    if constexpr (STATS) {   // the loop ended with a barrier: combine the four waves in a fixed order, one partial per workgroup
        for (int i = tid; i < Nc * 2; i += 256) {
            const int c2 = i;   // (channel, stat) pair; LDS rows are NTT*16 channels wide
            double v = 0.0;
#pragma unroll
            for (int w4 = 0; w4 < 4; ++w4) v += stat_lds[(size_t)w4 * NTT * 16 * 2 + c2];
            stat_part[(size_t)blockIdx.x * Nc * 2 + i] = v;
        }
    }
}

// ------------------------------------------------------------------ forward / dgrad without LDS: small volumes and strided layers
// The tiled kernel above needs thousands of 4x8x16 tiles to fill 512 workgroup slots and a stride of 1; the deep levels of
// Modified3DUNet (modified_3dunet.py:23-70: 64 -> 64 at 20x24x20, 128 -> 128 at 10x12x10, batch 1, and the four stride-2
// 3x3x3 layers between the levels), VoxResNet's stride-2 stem (cnn_model.py:49-81) and patch batches have neither, and ran on
// the direct (non-MFMA) kernels at 2 .. 13 TFLOP/s.  Here a WAVE is the unit: one M-tile of 16 voxels x NT N-tiles of 16 channels,
// the whole K = taps x Cin in one go, both operands straight from global memory (L2-resident at these sizes) — no LDS staging,
// no barriers, no tile waste.  Same operand order (weights, voxels), K permutation and epilogue as the tiled kernel; packed
// weights Wp[chunk16][tap][nt][lane][s] (pack_w_mfma_kernel, CK = 16).
//   MODE 0 (forward of a 3x3x3 / pad 1 / stride s layer; also the stride-1 data gradient = the forward of the flipped,
//           transposed weights on dY): the M-tile is 16 consecutive voxels of the flattened (n, od, oh, ow) OUTPUT index space;
//           tap (kd, kh, kw) reads input voxel (s*od - 1 + kd, ...), a constant element offset plus three range checks.
//   MODE 1 (data gradient of a stride-s layer, s > 1): dX[i] = sum over the taps with k = (i + 1) mod s of W[k] dY[(i + 1 - k)/s].
//           The M-tile is 16 input voxels of one (n, id, ih) row that share the residue of iw mod s, so the valid tap set —
//           1 .. 8 of the 27 — is wave-uniform: the tap loops just step by s and only the volume border is masked.
//   SPLIT: layers with fewer units than SIMDs give each unit to a whole workgroup: its four waves take the (kd, kh) pairs
//           round-robin and are summed through LDS in a fixed order (deterministic); otherwise a workgroup is four units.
// The loads of all kw taps of a (kd, kh) pair — up to 3 voxel fragments and 3 NT weight fragments per 16-channel chunk — are
// issued together before their MFMAs: a lone wave per SIMD then waits for L2 once per pair instead of once per MFMA group.
struct DirectGeom {
    int N, Di, Hi, Wi;   // the tensor the kernel READS (x, or dY for the gradients)
    int Do, Ho, Wo;      // the tensor it WRITES
    int s, Kc, in_ld, Nc, out_ld, NTT, gy, nmt, wbn;
};

template <typename T, int NT, int MODE, bool SPLIT>
__global__ void __launch_bounds__(256)
conv_mfma_direct_kernel(const T* __restrict__ in, const float* __restrict__ wp, const float* __restrict__ bias,
                        T* __restrict__ out, const DirectGeom q) {
    __shared__ float red[SPLIT ? 3 * NT * 256 : 1];
    const int lane = threadIdx.x & 63, li = lane & 15, kq = lane >> 4, wv = threadIdx.x >> 6;
    const int unit = SPLIT ? (int)blockIdx.x : (int)blockIdx.x * 4 + wv;
    if (unit >= q.nmt * q.gy) return;   // SPLIT: whole workgroup; otherwise the wave (no barrier follows)
    const int ntb = unit % q.gy, mt = unit / q.gy;
    const int s = q.s;

    // the lane's voxel: where it writes, and the read-side coordinates of its taps
    bool vok;
    int64_t out_off;           // element offset of the lane's output voxel
    int n, c_d, c_h, c_w;      // MODE 0: input coordinates of tap (0,0,0); MODE 1: the input-gradient voxel (id, ih, iw)
    if (MODE == 0) {
        const int64_t nvox = (int64_t)q.N * q.Do * q.Ho * q.Wo;
        const int64_t v = (int64_t)mt * 16 + li;
        vok = v < nvox;
        const int64_t vc = vok ? v : nvox - 1;
        const int ow = (int)(vc % q.Wo);
        int64_t t = vc / q.Wo;
        const int oh = (int)(t % q.Ho);
        t /= q.Ho;
        const int od = (int)(t % q.Do);
        n = (int)(t / q.Do);
        c_d = od * s - 1, c_h = oh * s - 1, c_w = ow * s - 1;
        out_off = vc * q.out_ld;
    } else {
        int r = mt / q.wbn;
        const int wb = mt % q.wbn;
        const int rw = r % s;
        r /= s;
        c_h = r % q.Ho;
        r /= q.Ho;
        c_d = r % q.Do;
        n = r / q.Do;
        c_w = rw + s * (wb * 16 + li);
        vok = c_w < q.Wo;
        out_off = ((((int64_t)n * q.Do + c_d) * q.Ho + c_h) * q.Wo + (vok ? c_w : 0)) * q.out_ld;
    }
    const T* const in_n = in + (int64_t)n * q.Di * q.Hi * q.Wi * q.in_ld;   // sample base: always readable (Kc >= 8)
    const int nchunks = (q.Kc + 15) / 16;
    const int nt0 = ntb * NT;
    const float* const wbase = wp + (size_t)nt0 * 256 + lane * 4;
    const size_t wtap = (size_t)q.NTT * 256, wchunk = (size_t)27 * q.NTT * 256;

    f32x4 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    // (kd, kh) pairs of this wave; MODE 1 starts at the residue tap and steps by the stride
    const int kd0 = MODE == 0 ? 0 : (c_d + 1) % s, kh0 = MODE == 0 ? 0 : (c_h + 1) % s, kstep = MODE == 0 ? 1 : s;
    int pair = 0;
#pragma unroll 1
    for (int kd = kd0; kd < 3; kd += kstep) {
        int rd;   // read-side d coordinate
        if (MODE == 0) rd = c_d + kd;
        else {
            const int nd = c_d + 1 - kd;
            rd = nd >= 0 ? nd / s : -1;
        }
        const bool okd = (unsigned)rd < (unsigned)q.Di;   // MODE 1: wave-uniform
#pragma unroll 1
        for (int kh = kh0; kh < 3; kh += kstep, ++pair) {
            if (SPLIT && (pair & 3) != wv) continue;
            int rh;
            if (MODE == 0) rh = c_h + kh;
            else {
                const int nh = c_h + 1 - kh;
                rh = nh >= 0 ? nh / s : -1;
            }
            const bool okh = okd && (unsigned)rh < (unsigned)q.Hi;
            if (MODE == 1 && !okh) continue;   // wave-uniform in MODE 1
            // the kw taps of this pair: source pointer (clamped to the sample base when masked), mask, weight tap
            const T* src[3];
            bool okw[3];
            int wtapi[3];
            int nkw = 0;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                int kw, rwc;
                bool have;
                if (MODE == 0) {
                    kw = j;
                    have = true;
                    rwc = c_w + kw;
                } else {
                    kw = (c_w + 1) % s + j * s;   // wave-uniform: c_w mod s is the M-tile's residue
                    have = kw < 3;
                    const int nw = c_w + 1 - kw;
                    rwc = nw >= 0 ? nw / s : -1;
                }
                const bool ok = have && vok && okh && (unsigned)rwc < (unsigned)q.Wi;
                okw[j] = ok;
                src[j] = ok ? in_n + (((int64_t)rd * q.Hi + rh) * q.Wi + rwc) * q.in_ld : in_n;
                const int tap = (kd * 3 + kh) * 3 + (have ? kw : 0);
                wtapi[j] = MODE == 0 ? tap : 26 - tap;   // MODE 1 reads the gradient image (pack_w_mfma_kernel dgrad = 1: 26 - tap)
                if (have) nkw = j + 1;
            }
#pragma unroll 1
            for (int ch = 0; ch < nchunks; ++ch) {
                const bool cok = ch * 16 + 4 * kq < q.Kc;   // Kc % 4 == 0 (host)
                float4 a[3];
                f32x4 b[3][NT];
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    if (j < nkw) {
                        a[j] = ldf4((okw[j] && cok) ? src[j] + ch * 16 + 4 * kq : in_n);
                        if (!(okw[j] && cok)) a[j] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            b[j][nt] = *reinterpret_cast<const f32x4*>(wbase + (size_t)wtapi[j] * wtap + (size_t)ch * wchunk + nt * 256);
                    }
                }
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    if (j < nkw) {
                        const float av[4] = {a[j].x, a[j].y, a[j].z, a[j].w};
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                            for (int s4 = 0; s4 < 4; ++s4)
                                acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[j][nt][s4], av[s4], acc[nt], 0, 0, 0);
                    }
                }
            }
        }
    }
    if (SPLIT) {   // waves 1..3 -> LDS, wave 0 adds them in a fixed order
        if (wv > 0) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[((wv - 1) * NT + nt) * 256 + r * 64 + lane] = acc[nt][r];
        }
        __syncthreads();
        if (wv > 0) return;
#pragma unroll
        for (int w = 0; w < 3; ++w)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[nt][r] += red[(w * NT + nt) * 256 + r * 64 + lane];
    }
    if (!vok) return;
    T* yv = out + out_off;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int co = (nt0 + nt) * 16 + 4 * kq;
        if (co >= q.Nc) continue;
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (bias) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (co + r < q.Nc) bv[r] = bias[co + r];
        }
        if (co + 3 < q.Nc && (q.out_ld & 3) == 0) {
            stf4(yv + co, make_float4(acc[nt][0] + bv[0], acc[nt][1] + bv[1], acc[nt][2] + bv[2], acc[nt][3] + bv[3]));
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (co + r < q.Nc) stf(yv + co + r, acc[nt][r] + bv[r]);
        }
    }
}

// ------------------------------------------------------------------ host side
struct MfmaFwdPlan {
    int CK, NT, NTT, gy, nchunks, tilesD, tilesH, tilesW, ntiles, grid;
    int small;   // served by the LDS-free kernel (conv_mfma_direct_kernel): 1 = fewer than 256 work units (or narrow / strided), 2 = fewer
                 // than one per workgroup slot (a preference: split operands and fused statistics stay on the tiled kernel)
    int narrow;  // ... or when the volume is narrower than a tile row (then also with BatchNorm statistics requested)
    size_t wp_floats, s_wp_floats, smem, stat_smem;   // packed-weight image of the tiled / the small-volume kernel
};

// Plan of the LDS-free kernel (conv_mfma_direct_kernel): 3x3x3, pad 1, dilation 1, one stride s for the three axes.
struct DirectPlan {
    DirectGeom q;
    int nt, mode, split, units;
    size_t wp_floats;
};

static bool direct_plan(const Mri3dConvGeom& g, bool dgrad, DirectPlan& p) {
    if (!(g.kd == 3 && g.kh == 3 && g.kw == 3 && g.pd == 1 && g.ph == 1 && g.pw == 1 && g.dd == 1 && g.dh == 1 && g.dw == 1 &&
          g.sd == g.sh && g.sh == g.sw && g.sd >= 1 && g.sd <= 3))
        return false;
    const int Kc = dgrad ? g.co : g.ci, Nc = dgrad ? g.ci : g.co;
    const int in_ld = dgrad ? g.y_ld : g.x_ld, out_ld = dgrad ? g.x_ld : g.y_ld;
    if (Kc % 4 != 0 || Kc < 8 || in_ld % 4 != 0 || Nc < 8) return false;   // 4-channel fragments; tiny outputs stay on the gather kernels
    DirectGeom& q = p.q;
    q.N = g.n;
    q.s = g.sd;
    p.mode = (dgrad && g.sd > 1) ? 1 : 0;
    if (!dgrad) {            // reads x (di..), writes y (dout..)
        q.Di = g.di, q.Hi = g.hi, q.Wi = g.wi, q.Do = g.dout, q.Ho = g.ho, q.Wo = g.wo;
    } else if (p.mode == 1) {   // reads dY (dout..), writes dX (di..)
        q.Di = g.dout, q.Hi = g.ho, q.Wi = g.wo, q.Do = g.di, q.Ho = g.hi, q.Wo = g.wi;
    } else {                 // stride 1: the gradient is the forward of the flipped weights on dY, same extents
        q.Di = g.dout, q.Hi = g.ho, q.Wi = g.wo, q.Do = g.di, q.Ho = g.hi, q.Wo = g.wi;
    }
    q.Kc = Kc, q.in_ld = in_ld, q.Nc = Nc, q.out_ld = out_ld;
    q.NTT = cdiv(Nc, 16);
    const int64_t nvox_out = (int64_t)q.N * q.Do * q.Ho * q.Wo;
    int64_t nmt;
    if (p.mode == 0) {
        q.wbn = 1;
        nmt = (nvox_out + 15) / 16;
    } else {
        q.wbn = cdiv(cdiv(q.Wo, q.s), 16);
        nmt = (int64_t)q.N * q.Do * q.Ho * q.s * q.wbn;
    }
    if (nmt * q.NTT > 0x3fffffff || nvox_out > 0x7fffffff || (int64_t)q.N * q.Di * q.Hi * q.Wi > 0x7fffffff) return false;
    q.nmt = (int)nmt;
    // enough waves for ~4 per SIMD where the layer allows it: narrower N-blocks when there are few M-tiles
    p.nt = (q.NTT % 4 == 0) ? 4 : ((q.NTT % 2 == 0) ? 2 : 1);
    while (p.nt > 1 && nmt * (q.NTT / p.nt) < 4096) p.nt >>= 1;
    q.gy = q.NTT / p.nt;
    p.units = q.nmt * q.gy;
    p.split = p.units < 1024 ? 1 : 0;   // fewer units than SIMDs: a workgroup per unit, its waves split the taps
    p.wp_floats = (size_t)cdiv(Kc, 16) * 27 * q.NTT * 256;   // [chunk16][tap][nt][lane][s]
    return true;
}

template <typename T>
static void launch_direct(const DirectPlan& p, const T* in, const float* wp, const float* bias, T* out, hipStream_t s) {
    const int grid = p.split ? p.units : cdiv(p.units, 4);
#define MRI3D_DIRECT_CASE(NTv, MODEv, SPv)                                                                            \
    if (p.nt == NTv && p.mode == MODEv && p.split == SPv)                                                             \
        hipLaunchKernelGGL((conv_mfma_direct_kernel<T, NTv, MODEv, (SPv != 0)>), dim3(grid), dim3(256), 0, s, in, wp, bias, out, p.q);
#define MRI3D_DIRECT_NT(MODEv, SPv) MRI3D_DIRECT_CASE(1, MODEv, SPv) MRI3D_DIRECT_CASE(2, MODEv, SPv) MRI3D_DIRECT_CASE(4, MODEv, SPv)
    MRI3D_DIRECT_NT(0, 0)
    MRI3D_DIRECT_NT(0, 1)
    MRI3D_DIRECT_NT(1, 0)
    MRI3D_DIRECT_NT(1, 1)
#undef MRI3D_DIRECT_NT
#undef MRI3D_DIRECT_CASE
}

// strided layers: served by the LDS-free kernel only
static bool direct_only(const Mri3dConvGeom& g) { return g.sd > 1 || g.sh > 1 || g.sw > 1; }

static bool mfma_fwd_plan(const Mri3dConvGeom& g, bool dgrad, MfmaFwdPlan& p) {
    if (!(g.kd == 3 && g.kh == 3 && g.kw == 3 && g.sd == 1 && g.sh == 1 && g.sw == 1 && g.pd == 1 && g.ph == 1 &&
          g.pw == 1 && g.dd == 1 && g.dh == 1 && g.dw == 1))
        return false;
    const bool bf = g.dtype == MRI3D_BF16;
    const int Kc = dgrad ? g.co : g.ci, Nc = dgrad ? g.ci : g.co;
    const int in_ld = dgrad ? g.y_ld : g.x_ld;
    if (Kc % 8 != 0 || in_ld % (bf ? 8 : 4) != 0) return false;   // 16-byte staging pieces
    if (Nc < 8) return false;  // tiny outputs (e.g. 16->2) stay on the direct kernel
    p.CK = bf ? 16 : 8;
    p.NTT = cdiv(Nc, 16);
    // NT (16-channel N-tiles per wave) is capped by registers: 8*NT accumulators + 2-deep A / 3-deep B rings + the staging
    // ring (NT <= 2).  Wider outputs are split over gy passes of the same tile.
    p.NT = (p.NTT % 2 == 0) ? 2 : 1;
    p.gy = p.NTT / p.NT;
    p.nchunks = cdiv(Kc, p.CK);
    p.tilesD = cdiv(g.di, TD);
    p.tilesH = cdiv(g.hi, TH);
    p.tilesW = cdiv(g.wi, TW);
    int64_t nt = (int64_t)g.n * p.tilesD * p.tilesH * p.tilesW;
    if (nt > 0x7fffffff) return false;
    p.ntiles = (int)nt;
    p.wp_floats = (size_t)p.nchunks * 14 * p.NTT * 256;   // 1 KiB per (chunk, tap group, N-tile)
    p.smem = (size_t)2 * kStg * 256 * 16;
    p.stat_smem = (size_t)4 * p.NTT * 16 * 2 * sizeof(double);   // per-wave float64 statistics of the STATS variant
    int64_t st = (int64_t)p.ntiles * p.gy;  // (spatial tile, n-tile block) work units
    if (st > 0x7fffffff) return false;
    p.grid = (int)std::min<int64_t>(st, 512);  // 2 resident workgroups per CU x 256 CUs
    // Small volumes (fp32): no more work units than workgroup slots — the wave-per-M-tile kernel fills the chip instead.  Its
    // operands come from L2 / the Infinity Cache, so it is only used while the input is small (<= 32 MB).
    p.small = 0;
    p.s_wp_floats = 0;
    const int64_t nvox = (int64_t)g.n * g.di * g.hi * g.wi;
    DirectPlan dp;
#ifndef MRI3D_SMALL_UNITS
#define MRI3D_SMALL_UNITS 512   // one work unit per workgroup slot or fewer.  Measured (tools/small_units_ab.sh, variants -DMRI3D_SMALL_UNITS=N): 256 -> 512 moves 32 -> 32
#endif                          // @ 40x48x40 x 2 from 64 to 72 and 64 -> 64 @ 40x48x40 from 68 to 80 TFLOP/s; 1024 loses on 32 -> 64 (85 -> 79)
    if (!bf && st < MRI3D_SMALL_UNITS && nvox * Kc * 4 <= ((int64_t)32 << 20) && direct_plan(g, dgrad, dp)) {   // bf16 tensors stay on the bf16 MFMA
        p.small = st < 256 ? 1 : 2;   // 2: a preference only — split operands and fused statistics still take the tiled kernel
        p.s_wp_floats = dp.wp_floats;
    }
    // Volumes at most half a tile row wide (the 8^3 level of the patch CNN, cnn_model.py:104-175, batch 512): a 16-voxel tile row
    // would be half padding.  The LDS-free kernel's M-tiles are 16 consecutive voxels of the flattened index space (two rows of
    // eight), nothing is wasted: 64 -> 64 @ 8^3 x 512 forward 61 -> 90, data gradient 65 -> 97 TFLOP/s.  (No fused BatchNorm
    // statistics there: conv_mfma_fwd_stat_blocks() answers 0 and the statistics pass reads the small output once.)
    p.narrow = 0;
    if (!bf && g.wi <= 8 && direct_plan(g, dgrad, dp)) {
        p.small = 1;
        p.narrow = 1;
        p.s_wp_floats = dp.wp_floats;
    }
    return true;
}

// second tensor of a split operand (conv over cat((x, x2), channels) / its data gradient written to two tensors): channels
// >= split live in `second` (pitch second_ld); second == nullptr: none
struct ConvSplit { const void* second; int split, second_ld; };

static int run_mfma_fwd(const Mri3dConvGeom& g, bool dgrad, const void* in_v, const float* w, const float* bias,
                        void* out_v, void* ws, size_t ws_bytes, hipStream_t s, double* stat_part = nullptr,
                        ConvSplit sp = ConvSplit{nullptr, 0, 0}) {
    MfmaFwdPlan p;
    DirectPlan dp;
    const bool strided = direct_only(g);
    // the layers the marching kernel is faster on (bf16 tensors on a chip-filling grid): conv_march.hip
    if (!strided && conv_march_takes(g, dgrad, stat_part != nullptr, false) &&
        (sp.second == nullptr || (sp.split % 16 == 0 && sp.second_ld % 8 == 0)))
        return conv_march_run(g, dgrad, false, in_v, w, bias, out_v, ws, ws_bytes, s, stat_part, sp.second, sp.split, sp.second_ld);
    if (strided) {
        MRI3D_REQUIRE(direct_plan(g, dgrad, dp) && stat_part == nullptr, MRI3D_ENOTSUP, "conv3d(mfma): unsupported strided geometry");
        p.small = 1;
        p.wp_floats = 0;
        p.s_wp_floats = dp.wp_floats;
    } else {
        MRI3D_REQUIRE(mfma_fwd_plan(g, dgrad, p), MRI3D_ENOTSUP, "conv3d(mfma): unsupported geometry");
    }
    const size_t need = std::max(p.wp_floats, p.s_wp_floats) * sizeof(float);
    MRI3D_REQUIRE(ws && ws_bytes >= need, MRI3D_EWORKSPACE, "conv3d(mfma): workspace %zu < %zu", ws_bytes, need);
    MRI3D_REQUIRE(((reinterpret_cast<uintptr_t>(in_v) | reinterpret_cast<uintptr_t>(out_v) | reinterpret_cast<uintptr_t>(ws)) & 15) == 0,
                  MRI3D_EINVAL, "conv3d(mfma): input/output/workspace must be 16-byte aligned");
    float* wp = static_cast<float*>(ws);
    const int Kc = dgrad ? g.co : g.ci, Nc = dgrad ? g.ci : g.co;
    const int in_ld = dgrad ? g.y_ld : g.x_ld, out_ld = dgrad ? g.x_ld : g.y_ld;
    int total = (int)p.wp_floats;
    MRI3D_REQUIRE(sp.second == nullptr || (p.small != 1 && !strided), MRI3D_ENOTSUP, "conv3d(mfma): split operands need the tiled kernel");
    if (p.small && stat_part == nullptr && sp.second == nullptr) {
        if (!strided) MRI3D_REQUIRE(direct_plan(g, dgrad, dp), MRI3D_ENOTSUP, "conv3d(mfma): unsupported geometry");
        const int stotal = (int)dp.wp_floats;
        hipLaunchKernelGGL(pack_w_mfma_kernel, dim3(std::min(cdiv(stotal, 256), 2048)), dim3(256), 0, s, w, wp, g.co, g.ci,
                           dgrad ? 1 : 0, 16, dp.q.NTT, cdiv(Kc, 16));
        MRI3D_DISPATCH_DTYPE(g.dtype, T, { launch_direct<T>(dp, (const T*)in_v, wp, bias, (T*)out_v, s); });
        return check_launch(dgrad ? "conv3d_dgrad(mfma direct)" : "conv3d_fwd(mfma direct)");
    }
    const bool n8 = Nc == 8 && stat_part == nullptr;   // exactly 8 output channels: the row-paired variant (10 groups per chunk)
    if (n8) {
        const int ptotal = p.nchunks * 10 * 256;   // 16-byte fragments x 64 lanes, in 4-byte units
        if (g.dtype == MRI3D_BF16)
            hipLaunchKernelGGL(pack_w_mfma_n8_kernel<bf16_t>, dim3(std::min(cdiv(2 * ptotal, 256), 2048)), dim3(256), 0, s, w,
                               reinterpret_cast<bf16_t*>(wp), g.co, g.ci, dgrad ? 1 : 0, p.nchunks);
        else
            hipLaunchKernelGGL(pack_w_mfma_n8_kernel<float>, dim3(std::min(cdiv(ptotal, 256), 2048)), dim3(256), 0, s, w, wp, g.co,
                               g.ci, dgrad ? 1 : 0, p.nchunks);
    } else if (g.dtype == MRI3D_BF16)   // same image size in bytes: 256 floats == 512 bf16 per (chunk, tg, nt)
        hipLaunchKernelGGL(pack_w_mfma_bf16_kernel, dim3(std::min(cdiv(2 * total, 256), 2048)), dim3(256), 0, s, w,
                           reinterpret_cast<bf16_t*>(wp), g.co, g.ci, dgrad ? 1 : 0, p.NTT, p.nchunks);
    else
        hipLaunchKernelGGL(pack_w_mfma_kernel, dim3(std::min(cdiv(total, 256), 2048)), dim3(256), 0, s, w, wp, g.co, g.ci,
                           dgrad ? 1 : 0, 8, p.NTT, p.nchunks);
    const int st = p.ntiles * p.gy;
    const size_t smem = p.smem + (stat_part ? p.stat_smem : 0);
    constexpr int kMaxSmem = 2 * kStg * 256 * 16 + 4 * 8 * 16 * 2 * 8;   // two halo buffers + float64 statistics of up to 128 channels
    MRI3D_REQUIRE(smem <= (size_t)kMaxSmem, MRI3D_ENOTSUP, "conv3d(mfma): too many output channels for fused statistics");
    // forward: the split is on the input (K) side; data gradient: on the output (N) side
    const void* x2 = dgrad ? nullptr : sp.second;
    void* y2 = dgrad ? const_cast<void*>(sp.second) : nullptr;
    const int x2_ld = dgrad ? 0 : sp.second_ld, ksplit = dgrad ? 0 : sp.split, y2_ld = dgrad ? sp.second_ld : 0, nsplit = dgrad ? sp.split : 0;
#define MRI3D_FWD2_CASE(NTv, STv, N8v)                                                                                \
    if (p.NT == NTv && (stat_part != nullptr) == STv && n8 == N8v) {                                                  \
        auto kern = conv_mfma_fwd2_kernel<T, NTv, STv, N8v>;                                                          \
        static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                       \
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, kMaxSmem);     \
        (void)attr;   /* once per kernel, not per launch */                                                          \
        hipLaunchKernelGGL(kern, dim3(p.grid), dim3(256), smem, s, (const T*)in_v, wp, bias, (T*)out_v, g.n, g.di,    \
                           g.hi, g.wi, Kc, in_ld, Nc, out_ld, p.NTT, p.gy, p.tilesD, p.tilesH, p.tilesW, st,         \
                           stat_part, (const T*)x2, x2_ld, ksplit, (T*)y2, y2_ld, nsplit);                           \
    }
    MRI3D_DISPATCH_DTYPE(g.dtype, T, {
        MRI3D_FWD2_CASE(1, false, false)
        MRI3D_FWD2_CASE(2, false, false)
        MRI3D_FWD2_CASE(1, true, false)
        MRI3D_FWD2_CASE(2, true, false)
        MRI3D_FWD2_CASE(1, false, true)
    });
#undef MRI3D_FWD2_CASE
    return check_launch(dgrad ? "conv3d_dgrad(mfma)" : "conv3d_fwd(mfma)");
}

int conv_mfma_fwd(const Mri3dConvGeom& g, const void* x, const float* w, const float* bias, void* y, void* ws,
                  size_t ws_bytes, hipStream_t s) {
    return run_mfma_fwd(g, false, x, w, bias, y, ws, ws_bytes, s);
}

// number of per-workgroup statistics partials the forward kernel writes for this geometry (0: not served by the MFMA path)
int conv_mfma_fwd_stat_blocks(const Mri3dConvGeom& g) {
    if (!direct_only(g) && conv_march_takes(g, false, true, false)) return conv_march_stat_blocks(g, false);
    MfmaFwdPlan p;
    if (!mfma_fwd_plan(g, false, p) || p.NTT > 8 || p.narrow) return 0;   // LDS statistics slots for up to 128 output channels
    return p.grid;
}

int conv_mfma_fwd_stats(const Mri3dConvGeom& g, const void* x, const float* w, const float* bias, void* y, double* stat_part,
                        void* ws, size_t ws_bytes, hipStream_t s) {
    MRI3D_REQUIRE(stat_part != nullptr, MRI3D_EINVAL, "conv3d_fwd_stats: null partial buffer");
    return run_mfma_fwd(g, false, x, w, bias, y, ws, ws_bytes, s, stat_part);
}

int conv_mfma_dgrad(const Mri3dConvGeom& g, const void* dy, const float* w, const float* bias, void* dx, void* ws,
                    size_t ws_bytes, hipStream_t s) {
    return run_mfma_fwd(g, true, dy, w, bias, dx, ws, ws_bytes, s);
}

// ---- split operands (conv over cat((x, x2), channels)): served by the tiled forward kernel and the transposed-tile weight
// gradient kernels only; `split` and the second tensor's channel count must be multiples of 16 (bf16 weight gradient: 8 for the
// second), the second tensor 16-byte aligned with a pitch like the first's
int conv_mfma_fwd_cat(const Mri3dConvGeom& g, const void* x, const void* x2, int split, int x2_ld, const float* w, const float* bias,
                      void* y, double* stat_part, void* ws, size_t ws_bytes, hipStream_t s) {
    return run_mfma_fwd(g, false, x, w, bias, y, ws, ws_bytes, s, stat_part, ConvSplit{x2, split, x2_ld});
}

int conv_mfma_dgrad_cat(const Mri3dConvGeom& g, const void* dy, const float* w, void* dx, void* dx2, int split, int dx2_ld, void* ws,
                        size_t ws_bytes, hipStream_t s) {
    return run_mfma_fwd(g, true, dy, w, nullptr, dx, ws, ws_bytes, s, nullptr, ConvSplit{dx2, split, dx2_ld});
}

// ================================================================== weight gradient
// GEMM view: dW(tap, ci; co) = sum over voxels.  One MFMA 16x16x4 takes A = X[4 voxels][16 rows] and B = dY[4 voxels][16 co]:
//   lane l supplies A[row = l&15][k = l>>4] and B[k = l>>4][col = l&15]; the 4 voxels of a k-step are consecutive in W.
//   rows of an M-tile:  CK=16: 16 input channels of one tap        (27 tap groups)
//                       CK= 8: 2 taps x 8 channels                 (14 tap groups, tap 27 = padding)
//                       CK= 1: 16 taps x the single input channel  ( 2 tap groups, taps 27..31 = padding)
// A workgroup (4 waves) is persistent over a contiguous range of 2x8x16-voxel tiles; it stages the X halo chunk and the
// dY tile (16 output channels) in LDS, each wave sweeps one quarter of the tile's voxels and keeps ALL tap groups of
// its (ci-tile, co-tile) pair in registers (27 x 4 VGPRs), so X and dY are read from LDS once per MFMA and from HBM/L2
// once per tile.  dbias rides along as one more accumulator fed with A = 1.  Partials are combined across the 4 waves
// through LDS in a fixed order, written once per workgroup, and summed by wgrad_mfma_reduce_kernel in a fixed order
// (deterministic: no float atomics).
constexpr int WTD = 2, WTH = 8, WTW = 16;
constexpr int WHD = WTD + 2, WHH = WTH + 2, WHW = WTW + 2;
constexpr int WHVOX = WHD * WHH * WHW;   // 720
constexpr int WVOX = WTD * WTH * WTW;    // 256

__host__ __device__ constexpr int wg_tap_groups(int CK) { return CK == 16 ? 27 : (CK == 8 ? 14 : 2); }

template <int CK>
__device__ __forceinline__ int wg_tap_offset(int tap) {  // halo-voxel offset of a tap (clamped to tap 26 for padding)
    const int t = tap < 27 ? tap : 26;
    return ((t / 9) * WHH + (t / 3) % 3) * WHW + t % 3;
}

// T = storage type of x / dy (bf16 tensors are widened to fp32 when they are staged: the MFMA arithmetic is fp32 in
// every wgrad kernel below)
template <typename T, int CK, bool BIAS>
__global__ void __launch_bounds__(256, 2)
conv_mfma_wgrad_kernel(const T* __restrict__ x, const T* __restrict__ dy, float* __restrict__ part, int N, int D,
                       int H, int W, int Ci, int x_ld, int Co, int y_ld, int tilesD, int tilesH, int tilesW, int ntiles) {
    constexpr int TG = wg_tap_groups(CK);
    constexpr int TGA = TG + (BIAS ? 1 : 0);
    constexpr int CP = CK;           // X tile voxel pitch (floats)
    constexpr int XV = CK >= 4 ? 4 : 1;  // staging vector width
    constexpr int XQ = CK / XV;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* xs = lds;                              // [WHVOX][CP]
    float* dys = lds + ((WHVOX * CP + 3) & ~3);   // [WVOX][16]

    const int cit = blockIdx.y, cob = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int li = lane & 15, kq = lane >> 4;
    const int dsel = wv >> 1, hsel = wv & 1;

    f32x4 acc[TGA];
#pragma unroll
    for (int t = 0; t < TGA; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

    // per-lane A offsets that do not depend on the voxel
    int lane_aoff[TG];
#pragma unroll
    for (int tg = 0; tg < TG; ++tg) {
        if (CK == 16) lane_aoff[tg] = wg_tap_offset<CK>(tg) * CP + li;
        else if (CK == 8) lane_aoff[tg] = wg_tap_offset<CK>(2 * tg + (li >> 3)) * CP + (li & 7);
        else lane_aoff[tg] = wg_tap_offset<CK>(16 * tg + li);
    }

    const int P = gridDim.x;
    const int t_lo = (int)(((int64_t)ntiles * blockIdx.x) / P), t_hi = (int)(((int64_t)ntiles * (blockIdx.x + 1)) / P);
    for (int tile = t_lo; tile < t_hi; ++tile) {
        int tt = tile;
        const int tw = tt % tilesW;
        tt /= tilesW;
        const int th = tt % tilesH;
        tt /= tilesH;
        const int td = tt % tilesD;
        const int n = tt / tilesD;
        const int w0 = tw * WTW, h0 = th * WTH, d0 = td * WTD;
        const T* xn = x + (int64_t)n * D * H * W * x_ld + cit * CK;
        const T* dn = dy + (int64_t)n * D * H * W * y_ld + cob * 16;

        __syncthreads();
        for (int idx = tid; idx < WHVOX * XQ; idx += 256) {
            const int q = idx % XQ, v = idx / XQ;
            const int wx = v % WHW;
            const int t2 = v / WHW;
            const int hy = t2 % WHH, dz = t2 / WHH;
            const int gd = d0 - 1 + dz, gh = h0 - 1 + hy, gw = w0 - 1 + wx;
            const bool ok = (unsigned)gd < (unsigned)D && (unsigned)gh < (unsigned)H && (unsigned)gw < (unsigned)W;
            const T* src = xn + (((int64_t)gd * H + gh) * W + gw) * x_ld + XV * q;
            if (XV == 4) {
                float4 val = ok ? ldf4(src) : make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<float4*>(xs + v * CP + 4 * q) = val;
            } else {
                xs[v * CP + q] = ok ? ldf(src) : 0.f;
            }
        }
        for (int idx = tid; idx < WVOX * 4; idx += 256) {
            const int q = idx & 3, v = idx >> 2;
            const int wx = v % WTW;
            const int t2 = v / WTW;
            const int hy = t2 % WTH, dz = t2 / WTH;
            const int gd = d0 + dz, gh = h0 + hy, gw = w0 + wx;
            const bool ok = gd < D && gh < H && gw < W;
            const T* src = dn + (((int64_t)gd * H + gh) * W + gw) * y_ld + 4 * q;
            float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
            const int cbase = cob * 16 + 4 * q;
            if (ok) {
                if (cbase + 3 < Co && (y_ld & 3) == 0) {
                    val = ldf4(src);
                } else {
                    if (cbase + 0 < Co) val.x = ldf(src);
                    if (cbase + 1 < Co) val.y = ldf(src + 1);
                    if (cbase + 2 < Co) val.z = ldf(src + 2);
                    if (cbase + 3 < Co) val.w = ldf(src + 3);
                }
            }
            *reinterpret_cast<float4*>(dys + v * 16 + 4 * q) = val;
        }
        __syncthreads();

#pragma unroll 1
        for (int hr = 0; hr < 4; ++hr) {
            const int hy = hsel * 4 + hr;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int wx = ks * 4 + kq;
                const float b = dys[((dsel * WTH + hy) * WTW + wx) * 16 + li];
                const float* abase = xs + ((dsel * WHH + hy) * WHW + wx) * CP;
#pragma unroll
                for (int tg = 0; tg < TG; ++tg) {
                    const float a = abase[lane_aoff[tg]];
                    acc[tg] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[tg], 0, 0, 0);
                }
                if (BIAS) acc[TG] = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, b, acc[TG], 0, 0, 0);
            }
        }
    }

    // combine the 4 waves in a fixed order through LDS, then one partial per workgroup
    __syncthreads();
    float* red = lds;  // [TGA][256]
    for (int w = 0; w < 4; ++w) {
        if (wv == w) {
#pragma unroll
            for (int t = 0; t < TGA; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int o = t * 256 + (4 * kq + r) * 16 + li;
                    red[o] = (w == 0) ? acc[t][r] : red[o] + acc[t][r];
                }
        }
        __syncthreads();
    }
    float* out = part + (((size_t)blockIdx.x * gridDim.y + cit) * gridDim.z + cob) * (TGA * 256);
    for (int i = tid; i < TGA * 256; i += 256) out[i] = red[i];
}

// Tile -> persistent-workgroup map shared by the wgrad kernels (speed only; see conv_mfma_fwd2_kernel): workgroups that
// land on the same XCD (linear block id mod 8) take CONSECUTIVE tiles of that XCD's contiguous tile range round-robin.
struct TileWalk { int first, stride, count; };
__device__ __forceinline__ TileWalk tile_walk(int ntiles) {
    const int P = gridDim.x;
    const int NX = P < 8 ? P : 8;
    const int off = (int)(((int64_t)P * (blockIdx.y + gridDim.y * blockIdx.z)) % NX);
    const int grp = ((int)blockIdx.x + off) % NX;
    const int p0 = (grp - off + NX) % NX;                 // first blockIdx.x of this XCD group in this (y, z) row
    const int slot = ((int)blockIdx.x - p0) / NX, members = (P - p0 + NX - 1) / NX;
    const int r_lo = (int)(((int64_t)ntiles * grp) / NX), r_hi = (int)(((int64_t)ntiles * (grp + 1)) / NX);
    TileWalk w;
    w.first = r_lo + slot;
    w.stride = members;
    w.count = w.first < r_hi ? (r_hi - w.first + members - 1) / members : 0;
    return w;
}

// ------------------------------------------------------------------ weight gradient, version 3 (CK = 16 or 8)
// v1 spends 2.3 of 5.1 ms of the 48->16 layer staging tiles while no MFMA runs (ablation: compute-only 130 TFLOP/s);
// the two resident workgroups of a CU run in lock-step, so nothing hides it.  v3 keeps v1's compute (all tap-group
// accumulators in registers, conflict-free 16-row M-tiles) and splits the staging T14-style: the NEXT tile's 16-byte
// pieces (12 of X + 4 of dY per lane) are fetched into registers before the current tile's 432 MFMAs per wave and
// written to LDS after them, so HBM/L2 latency is covered by MFMA work and only the LDS write pass stays exposed.
template <typename T, int CK, bool BIAS>
__global__ void __launch_bounds__(256, 2)
conv_mfma_wgrad3_kernel(const T* __restrict__ x, const T* __restrict__ dy, float* __restrict__ part, int N, int D,
                        int H, int W, int Ci, int x_ld, int Co, int y_ld, int tilesD, int tilesH, int tilesW, int ntiles) {
    constexpr int TG = wg_tap_groups(CK);
    constexpr int TGA = TG + (BIAS ? 1 : 0);
    constexpr int CP = CK;
    constexpr int XQ = CK / 4;
    constexpr int NPX = (WHVOX * XQ + 255) / 256;  // X pieces per lane (12 for CK=16, 6 for CK=8)
    constexpr int NPY = WVOX * 4 / 256;            // dY pieces per lane (4)
    constexpr int XBUF = NPX * 256 * 4;            // floats (padded to whole pieces)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* xs = lds;           // [WHVOX][CP] (+ padding)
    float* dys = lds + XBUF;   // [WVOX][16]

    const int cit = blockIdx.y, cob = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int li = lane & 15, kq = lane >> 4;
    const int dsel = wv >> 1, hsel = wv & 1;

    f32x4 acc[TGA];
#pragma unroll
    for (int t = 0; t < TGA; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

    int lane_aoff[TG];
#pragma unroll
    for (int tg = 0; tg < TG; ++tg) {
        if (CK == 16) lane_aoff[tg] = wg_tap_offset<CK>(tg) * CP + li;
        else lane_aoff[tg] = wg_tap_offset<CK>(2 * tg + (li >> 3)) * CP + (li & 7);
    }

    struct Tile { int n, d0, h0, w0; };
    auto decode = [&](int tile) -> Tile {
        Tile r;
        r.w0 = (tile % tilesW) * WTW;
        tile /= tilesW;
        r.h0 = (tile % tilesH) * WTH;
        tile /= tilesH;
        r.d0 = (tile % tilesD) * WTD;
        r.n = tile / tilesD;
        return r;
    };
    float4 px[NPX], py[NPY];
    unsigned xok = 0, yok = 0;
    // unconditional (clamped) loads; out-of-volume pieces are zeroed when they are written to LDS
    auto load_tile = [&](const Tile& t) {
#pragma unroll
        for (int j = 0; j < NPX; ++j) {
            const int idx = j * 256 + tid;
            const int pv = idx < WHVOX * XQ ? idx : 0;
            const int q = pv % XQ, v = pv / XQ;
            const int wx = v % WHW, t2 = v / WHW;
            const int gd = t.d0 - 1 + t2 / WHH, gh = t.h0 - 1 + t2 % WHH, gw = t.w0 - 1 + wx;
            const bool ok = (unsigned)gd < (unsigned)D && (unsigned)gh < (unsigned)H && (unsigned)gw < (unsigned)W;
            xok = ok ? (xok | (1u << j)) : (xok & ~(1u << j));
            const int cd = min(max(gd, 0), D - 1), chh = min(max(gh, 0), H - 1), cw = min(max(gw, 0), W - 1);
            px[j] = ldf4(x + ((((int64_t)t.n * D + cd) * H + chh) * W + cw) * x_ld + cit * CK +
                                                     4 * q);
        }
#pragma unroll
        for (int j = 0; j < NPY; ++j) {
            const int idx = j * 256 + tid;
            const int q = idx & 3, v = idx >> 2;
            const int wx = v % WTW, t2 = v / WTW;
            const int gd = t.d0 + t2 / WTH, gh = t.h0 + t2 % WTH, gw = t.w0 + wx;
            const int cb = cob * 16 + 4 * q;
            const bool ok = gd < D && gh < H && gw < W && cb < Co;   // host guarantees Co % 4 == 0
            yok = ok ? (yok | (1u << j)) : (yok & ~(1u << j));
            const int cd = min(gd, D - 1), chh = min(gh, H - 1), cw = min(gw, W - 1), cc = min(cb, Co - 4);
            py[j] = ldf4(dy + ((((int64_t)t.n * D + cd) * H + chh) * W + cw) * y_ld + cc);
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int j = 0; j < NPX; ++j) {
            const bool ok = (xok >> j) & 1u;
            float4 v2;
            v2.x = ok ? px[j].x : 0.f; v2.y = ok ? px[j].y : 0.f; v2.z = ok ? px[j].z : 0.f; v2.w = ok ? px[j].w : 0.f;
            *reinterpret_cast<float4*>(xs + (j * 256 + tid) * 4) = v2;
        }
#pragma unroll
        for (int j = 0; j < NPY; ++j) {
            const bool ok = (yok >> j) & 1u;
            float4 v2;
            v2.x = ok ? py[j].x : 0.f; v2.y = ok ? py[j].y : 0.f; v2.z = ok ? py[j].z : 0.f; v2.w = ok ? py[j].w : 0.f;
            *reinterpret_cast<float4*>(dys + (j * 256 + tid) * 4) = v2;
        }
    };

    const TileWalk tw = tile_walk(ntiles);
    if (tw.count > 0) {
        load_tile(decode(tw.first));
        store_tile();
        __syncthreads();
        for (int k = 0; k < tw.count; ++k) {
            const bool has_next = k + 1 < tw.count;
            if (has_next) load_tile(decode(tw.first + (k + 1) * tw.stride));   // global -> registers, in flight during the MFMAs
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
            for (int hr = 0; hr < 4; ++hr) {
                const int hy = hsel * 4 + hr;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const int wx = ks * 4 + kq;
                    const float b = dys[((dsel * WTH + hy) * WTW + wx) * 16 + li];
                    const float* abase = xs + ((dsel * WHH + hy) * WHW + wx) * CP;
#pragma unroll
                    for (int tg = 0; tg < TG; ++tg) {
                        const float a = abase[lane_aoff[tg]];
                        acc[tg] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[tg], 0, 0, 0);
                    }
                    if (BIAS) acc[TG] = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, b, acc[TG], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();                 // every wave is done reading this tile
            if (has_next) store_tile();      // registers -> LDS
            __syncthreads();
        }
    }

    // combine the 4 waves in a fixed order through LDS, then one partial per workgroup
    __syncthreads();
    float* red = lds;  // [TGA][256]
    for (int w = 0; w < 4; ++w) {
        if (wv == w) {
#pragma unroll
            for (int t = 0; t < TGA; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int o = t * 256 + (4 * kq + r) * 16 + li;
                    red[o] = (w == 0) ? acc[t][r] : red[o] + acc[t][r];
                }
        }
        __syncthreads();
    }
    float* out = part + (((size_t)blockIdx.x * gridDim.y + cit) * gridDim.z + cob) * (TGA * 256);
    for (int i = tid; i < TGA * 256; i += 256) out[i] = red[i];
}

// ------------------------------------------------------------------ weight gradient, version 4 (Cin % 16 == 0)
// v3's overlap without its register bill: a 2x4x16-voxel tile (X halo 27 KB + dY 8 KB) is small enough to double-buffer
// in LDS with two workgroups per CU, and its 9 pieces per lane fit in registers next to the 27 tap accumulators.
// Per tile every wave runs 2 h-rows x 4 k-steps x 27 MFMAs: the next tile's pieces are fetched during row 0 and written
// to the OTHER buffer during row 1 — no exposed staging pass and a single barrier per tile.
constexpr int V4TH = 4, V4HH = V4TH + 2;
constexpr int V4HVOX = WHD * V4HH * WHW;           // 4 x 6 x 18 = 432 halo voxels
constexpr int V4VOX = WTD * V4TH * WTW;            // 128 output voxels
constexpr int V4NPX = (V4HVOX * 4 + 255) / 256;    // 7 X pieces per lane
constexpr int V4NPY = V4VOX * 4 / 256;             // 2 dY pieces per lane
constexpr int V4XBUF = V4NPX * 256 * 4;            // floats
constexpr int V4YBUF = V4VOX * 16;                 // floats

__device__ __forceinline__ int v4_tap_offset(int tap) { return ((tap / 9) * V4HH + (tap / 3) % 3) * WHW + tap % 3; }

template <typename T, bool BIAS>
__global__ void __launch_bounds__(256, 2)
conv_mfma_wgrad4_kernel(const T* __restrict__ x, const T* __restrict__ dy, float* __restrict__ part, int N, int D,
                        int H, int W, int Ci, int x_ld, int Co, int y_ld, int tilesD, int tilesH, int tilesW, int ntiles) {
    constexpr int TG = 27, TGA = TG + (BIAS ? 1 : 0), CP = 16;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* xs = lds;                    // [2][V4XBUF]
    float* dys = lds + 2 * V4XBUF;      // [2][V4YBUF]

    const int cit = blockIdx.y, cob = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int li = lane & 15, kq = lane >> 4;
    const int dsel = wv >> 1, hsel = wv & 1;

    f32x4 acc[TGA];
#pragma unroll
    for (int t = 0; t < TGA; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

    struct Tile { int n, d0, h0, w0; };
    auto decode = [&](int tile) -> Tile {
        Tile r;
        r.w0 = (tile % tilesW) * WTW;
        tile /= tilesW;
        r.h0 = (tile % tilesH) * V4TH;
        tile /= tilesH;
        r.d0 = (tile % tilesD) * WTD;
        r.n = tile / tilesD;
        return r;
    };
    float4 px[V4NPX], py[V4NPY];
    unsigned xok = 0, yok = 0;
    // Per-lane piece offsets relative to the tile's halo origin, computed ONCE: for an interior tile (the common case) a
    // piece's address is a wave-uniform base plus this 32-bit offset, i.e. no per-tile coordinate arithmetic at all.
    // The two workgroups of a CU run in lock-step, so every VALU instruction spent on addressing is time the MFMA pipe
    // idles (ablation: loads+stores alone 1.09 ms, MFMA alone 3.35 ms, together 4.26 ms on the 48->16 layer).
    unsigned xrel[V4NPX], yrel[V4NPY];
#pragma unroll
    for (int j = 0; j < V4NPX; ++j) {
        const int idx = j * 256 + tid;
        const int pv = idx < V4HVOX * 4 ? idx : 0;
        const int q = pv & 3, v = pv >> 2;
        const int wx = v % WHW, t2 = v / WHW;
        xrel[j] = (unsigned)((((t2 / V4HH) * H + t2 % V4HH) * W + wx) * x_ld + 4 * q);
    }
#pragma unroll
    for (int j = 0; j < V4NPY; ++j) {
        const int idx = j * 256 + tid;
        const int q = idx & 3, v = idx >> 2;
        const int wx = v % WTW, t2 = v / WTW;
        yrel[j] = (unsigned)((((t2 / V4TH) * H + t2 % V4TH) * W + wx) * y_ld + 4 * q);
    }
    const bool co_full = cob * 16 + 16 <= Co;
    auto load_tile = [&](const Tile& t) {
        const bool interior = t.d0 >= 1 && t.d0 + WTD < D && t.h0 >= 1 && t.h0 + V4TH < H && t.w0 >= 1 && t.w0 + WTW < W &&
                              co_full;   // wave-uniform
        if (interior) {
            const T* xo = x + ((((int64_t)t.n * D + t.d0 - 1) * H + t.h0 - 1) * W + t.w0 - 1) * x_ld + cit * 16;
            const T* yo = dy + ((((int64_t)t.n * D + t.d0) * H + t.h0) * W + t.w0) * y_ld + cob * 16;
#pragma unroll
            for (int j = 0; j < V4NPX; ++j) px[j] = ldf4(xo + xrel[j]);
#pragma unroll
            for (int j = 0; j < V4NPY; ++j) py[j] = ldf4(yo + yrel[j]);
            xok = ~0u;
            yok = ~0u;
            return;
        }
        // border tiles: clamped addresses, zeroing happens at store time
#pragma unroll
        for (int j = 0; j < V4NPX; ++j) {
            const int idx = j * 256 + tid;
            const int pv = idx < V4HVOX * 4 ? idx : 0;
            const int q = pv & 3, v = pv >> 2;
            const int wx = v % WHW, t2 = v / WHW;
            const int gd = t.d0 - 1 + t2 / V4HH, gh = t.h0 - 1 + t2 % V4HH, gw = t.w0 - 1 + wx;
            const bool ok = (unsigned)gd < (unsigned)D && (unsigned)gh < (unsigned)H && (unsigned)gw < (unsigned)W;
            xok = ok ? (xok | (1u << j)) : (xok & ~(1u << j));
            const int cd = min(max(gd, 0), D - 1), chh = min(max(gh, 0), H - 1), cw = min(max(gw, 0), W - 1);
            px[j] = ldf4(x + ((((int64_t)t.n * D + cd) * H + chh) * W + cw) * x_ld + cit * 16 +
                                                     4 * q);
        }
#pragma unroll
        for (int j = 0; j < V4NPY; ++j) {
            const int idx = j * 256 + tid;
            const int q = idx & 3, v = idx >> 2;
            const int wx = v % WTW, t2 = v / WTW;
            const int gd = t.d0 + t2 / V4TH, gh = t.h0 + t2 % V4TH, gw = t.w0 + wx;
            const int cb = cob * 16 + 4 * q;
            const bool ok = gd < D && gh < H && gw < W && cb < Co;   // host guarantees Co % 4 == 0
            yok = ok ? (yok | (1u << j)) : (yok & ~(1u << j));
            const int cd = min(gd, D - 1), chh = min(gh, H - 1), cw = min(gw, W - 1), cc = min(cb, Co - 4);
            py[j] = ldf4(dy + ((((int64_t)t.n * D + cd) * H + chh) * W + cw) * y_ld + cc);
        }
    };
    auto store_tile = [&](float* xb, float* yb) {
        if ((xok & yok) == ~0u) {   // interior tile (wave-uniform): no masking
#pragma unroll
            for (int j = 0; j < V4NPX; ++j) *reinterpret_cast<float4*>(xb + (j * 256 + tid) * 4) = px[j];
#pragma unroll
            for (int j = 0; j < V4NPY; ++j) *reinterpret_cast<float4*>(yb + (j * 256 + tid) * 4) = py[j];
            return;
        }
#pragma unroll
        for (int j = 0; j < V4NPX; ++j) {
            const bool ok = (xok >> j) & 1u;
            float4 v2;
            v2.x = ok ? px[j].x : 0.f; v2.y = ok ? px[j].y : 0.f; v2.z = ok ? px[j].z : 0.f; v2.w = ok ? px[j].w : 0.f;
            *reinterpret_cast<float4*>(xb + (j * 256 + tid) * 4) = v2;
        }
#pragma unroll
        for (int j = 0; j < V4NPY; ++j) {
            const bool ok = (yok >> j) & 1u;
            float4 v2;
            v2.x = ok ? py[j].x : 0.f; v2.y = ok ? py[j].y : 0.f; v2.z = ok ? py[j].z : 0.f; v2.w = ok ? py[j].w : 0.f;
            *reinterpret_cast<float4*>(yb + (j * 256 + tid) * 4) = v2;
        }
    };
    auto row = [&](const float* xb, const float* yb, int hr) {
        const int hy = hsel * 2 + hr;
        // a REAL loop over the 4 k-steps (one basic block each): unrolled, hipcc hoists all four k-steps' LDS reads
        // (108 VGPRs) on top of the 112 accumulators + 36 staging registers and spills (78 vs 96 TFLOP/s measured); a
        // ping-pong prefetch of the next k-step's fragments (2 x 28 VGPRs) spills as well.
#pragma unroll 1
        for (int ks = 0; ks < 4; ++ks) {
            const int wx = ks * 4 + kq;
            const float b = yb[((dsel * V4TH + hy) * WTW + wx) * 16 + li];
            const float* abase = xb + ((dsel * V4HH + hy) * WHW + wx) * CP + li;
#pragma unroll
            for (int tg = 0; tg < TG; ++tg) {
                const float a = abase[v4_tap_offset(tg) * CP];
                acc[tg] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[tg], 0, 0, 0);
            }
            if (BIAS) acc[TG] = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, b, acc[TG], 0, 0, 0);
        }
    };

    const TileWalk tw = tile_walk(ntiles);
    if (tw.count > 0) {
        load_tile(decode(tw.first));
        store_tile(xs, dys);
        __syncthreads();
        for (int k = 0; k < tw.count; ++k) {
            const int cb = k & 1;
            const float* xb = xs + cb * V4XBUF;
            const float* yb = dys + cb * V4YBUF;
            const bool has_next = k + 1 < tw.count;
            if (has_next) load_tile(decode(tw.first + (k + 1) * tw.stride));        // global -> registers
            __builtin_amdgcn_sched_barrier(0);
            row(xb, yb, 0);                                   // 108 MFMAs per wave cover the loads
            if (has_next) store_tile(xs + (cb ^ 1) * V4XBUF, dys + (cb ^ 1) * V4YBUF);  // registers -> the OTHER buffer
            row(xb, yb, 1);
            __syncthreads();
        }
    }

    // combine the 4 waves in a fixed order through LDS, then one partial per workgroup
    __syncthreads();
    float* red = lds;  // [TGA][256]
    for (int w = 0; w < 4; ++w) {
        if (wv == w) {
#pragma unroll
            for (int t = 0; t < TGA; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int o = t * 256 + (4 * kq + r) * 16 + li;
                    red[o] = (w == 0) ? acc[t][r] : red[o] + acc[t][r];
                }
        }
        __syncthreads();
    }
    float* out = part + (((size_t)blockIdx.x * gridDim.y + cit) * gridDim.z + cob) * (TGA * 256);
    for (int i = tid; i < TGA * 256; i += 256) out[i] = red[i];
}

// ------------------------------------------------------------------ weight gradient, transposed tiles (bf16 MFMA and fp32 v6)
// Both kernels below keep the tile TRANSPOSED in LDS, [channel][voxel]: the MFMA sums over voxels, and a lane's operand is a run
// of consecutive voxels of one channel.  Tile = 2 x 6 rows of TW voxels (TW = 32 bf16 / 16 fp32); X needs its (kd, kh) halo
// rows (4 x 8 rows), dY its 12 output rows.
//
// The kw taps:  dW[kd,kh,kw] = sum_v X[v + kw - 1] dY[v]  =  sum_u X[u] dY[u + 1 - kw].  The shift is applied to dY, not to X:
// per output row the three fragments dY[u+1], dY[u], dY[u-1] are built ONCE (register selection / v_alignbyte from the aligned
// fragment and its two neighbour voxels) and every (kd, kh) then costs ONE aligned X read for three MFMAs.  (Round 1 shifted X:
// one aligned read plus two neighbour reads per (kd, kh), 3-way bank-conflicted — PMC: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE =
// 0.64 in the bf16 kernel, which was LDS-bound at 18 % MFMA busy.)  The sums are re-partitioned between W-neighbouring tiles —
// a tile now takes the products of ITS X voxels, with dY[w0-1] and dY[w0+TW] read from the neighbours (zero outside the volume)
// — so X has no W halo at all and dY has a one-voxel W halo; the total over tiles is unchanged.
//
// LDS image: a (row, channel) line is 64 bytes = four 16-byte slots; logical slot q of channel c sits at physical slot
// (q + 2*(c >> 3)) & 3.  The hardware serves a ds_read_b128 in the lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ...
// (MI355X_MICROARCH.md §LDS), i.e. channels 4..11 of a group read the NEXT k-group's slot; with unpadded 64-byte lines and the
// two-slot rotation of channels 8..15 all 16 lanes of every group hit different 4-bank slots (the 80-byte padded lines of round
// 1 cost 8 instead of 4 LDS cycles per read).
constexpr int BTD = 2, BTH = 6, BTW = 32;
constexpr int BHD = BTD + 2, BHH = BTH + 2;
constexpr int BXR = BHD * BHH;            // 32 X rows
constexpr int BYR = BTD * BTH;            // 12 dY rows
constexpr int DLS = 64;                   // bytes per (row, channel) line
constexpr int BXS = BXR * 16 * DLS;       // 32768 B  X
constexpr int BYS = BYR * 16 * DLS;       // 12288 B  dY
constexpr int BYH = BYR * 16 * 8;         //  1536 B  dY W-halo: per line {dword holding dY[w0-1], dword holding dY[w0+TW]}
// Sixteen zero bytes in device memory: what an out-of-volume 16-byte piece of a border tile loads.  Selecting the ADDRESS
// (piece or zeros) instead of the loaded VALUE (`ok ? loaded : 0`) keeps the staging wait-free: a select on loaded data makes
// hipcc wait for the load on the spot, i.e. the full memory latency in front of the tile's MFMAs.
__device__ const float g_zero16[4] = {0.f, 0.f, 0.f, 0.f};

__device__ __forceinline__ int rot_slot(int q, int c) { return (q + 2 * ((c >> 3) & 1)) & 3; }

// eight voxels x eight channels (v[j] = the 16-byte channel vector of voxel j) -> out[c] = the 8 voxels of channel c
__device__ __forceinline__ void transpose8x8_bf16(const uint4 (&v)[8], uint4 (&out)[8]) {
    const unsigned* vw = reinterpret_cast<const unsigned*>(v);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int cd = c >> 1;
        const unsigned sel = (c & 1) ? 0x07060302u : 0x05040100u;   // high / low halves of (S1 = even voxel, S0 = odd voxel)
        out[c].x = __builtin_amdgcn_perm(vw[1 * 4 + cd], vw[0 * 4 + cd], sel);
        out[c].y = __builtin_amdgcn_perm(vw[3 * 4 + cd], vw[2 * 4 + cd], sel);
        out[c].z = __builtin_amdgcn_perm(vw[5 * 4 + cd], vw[4 * 4 + cd], sel);
        out[c].w = __builtin_amdgcn_perm(vw[7 * 4 + cd], vw[6 * 4 + cd], sel);
    }
}

// 16-byte load of bf16 data through an explicitly GLOBAL pointer (global_load_dwordx4: vmcnt only, never lgkmcnt)
__device__ __forceinline__ uint4 ldg4u(const bf16_t* p) {
    typedef unsigned gu32x4 __attribute__((ext_vector_type(4)));
    const gu32x4 v = *(const __attribute__((address_space(1))) gu32x4*)p;
    return make_uint4(v[0], v[1], v[2], v[3]);
}

// v_mfma_f32_16x16x32_bf16 sums over K = 32 VOXELS with 8 consecutive k per lane.  Staging: each lane loads 8 consecutive
// voxels x 8 channels (8 x 16 B) and transposes them in registers (v_perm) into eight 16-byte LDS writes.
// Accumulators: 27 taps x (16 ci x 16 co) per wave (+1 for dbias, fed with A = 1).
template <bool BIAS>
__global__ void __launch_bounds__(256, 2)
conv_mfma_wgrad_bf16_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy, float* __restrict__ part, int N,
                            int D, int H, int W, int Ci, int x_ld, int Co, int y_ld, int tilesD, int tilesH, int tilesW,
                            int ntiles, const bf16_t* __restrict__ x2, int x2_ld, int ksplit) {
    constexpr int TG = 27, TGA = TG + (BIAS ? 1 : 0);
    // conv over cat((x, x2)): this workgroup's 16-channel ci-tile lives in ONE of the two tensors — rebind x / pitch / channel origin
    int xc0 = (int)blockIdx.y * 16, xcn = Ci;   // first channel of the tile inside its tensor, channels of that tensor
    if (x2 != nullptr) {
        if (xc0 >= ksplit) { x = x2; x_ld = x2_ld; xc0 -= ksplit; xcn = Ci - ksplit; }
        else xcn = ksplit;
    }
    extern __shared__ __attribute__((aligned(16))) float lds[];
    char* xs = reinterpret_cast<char*>(lds);
    char* ys = xs + BXS;
    char* yh = ys + BYS;

    const int cit = blockIdx.y, cob = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int li = lane & 15, kq = lane >> 4;

    f32x4 acc[TGA];
#pragma unroll
    for (int t = 0; t < TGA; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

    // staging roles (fixed per lane)
    const int s_half = tid & 1, s_wg = (tid >> 1) & 3, s_row = tid >> 3;     // X: 32 rows x 4 w-groups x 2 channel halves
    const int h_half = tid & 1, h_side = (tid >> 1) & 1;   // dY halo voxels (lanes 128 .. 175)
    const uint4 zero4 = make_uint4(0u, 0u, 0u, 0u);
    // per-lane operand addresses
    const int orow0 = wv * (BYR / 4);   // first of the wave's three output rows (same d-plane, consecutive h)
    const char* const xrow0 = xs + (((orow0 / BTH) * BHH + orow0 % BTH) * 16 + li) * DLS + 16 * rot_slot(kq, li);
    const int yline0 = orow0 * 16 + li;
    const char* const yrow0 = ys + yline0 * DLS + 16 * rot_slot(kq, li);
    // the dword holding the voxel before / after the lane's eight: last dword of the previous / first dword of the next k-group's
    // slot, or the W-halo entry of the line (k-groups 0 and 3)
    const char* const ypl0 = kq == 0 ? yh + yline0 * 8 : ys + yline0 * DLS + 16 * rot_slot(kq - 1, li) + 12;
    const char* const ynr0 = kq == 3 ? yh + yline0 * 8 + 4 : ys + yline0 * DLS + 16 * rot_slot(kq + 1, li);
    const int pl_step = kq == 0 ? 16 * 8 : 16 * DLS, nr_step = kq == 3 ? 16 * 8 : 16 * DLS;
    bf16x8_t ones;
#pragma unroll
    for (int i = 0; i < 8; ++i) ones[i] = (bf16_t)1.0f;

    // The next tile's pieces are fetched into registers while the current tile is multiplied and transposed / written to the single
    // LDS tile between two barriers after it (as in wgrad6): the HBM latency of a tile is no longer exposed in front of its MFMAs
    // (round 1 staged synchronously: 18 % MFMA busy).
    uint4 vx[8], vy[8];
    uint4& vh = vy[0];   // the halo lanes (128 .. 175) stage no dY rows: their one piece shares a register with them
    const bool is_y = tid < BYR * 8, is_h = tid >= 128 && tid < 128 + BYR * 4;
    const int hy_row = (tid - 128) >> 2;   // dY halo row of lanes 128 .. 175
    // per-lane byte-free element offsets from the tile's origin voxels (X: (d0-1, h0-1, w0); dY: (d0, h0, w0); halo: (d0, h0, w0-1))
    const unsigned xrel = (unsigned)((((s_row / BHH) * H + s_row % BHH) * W + 8 * s_wg) * x_ld + 8 * s_half);
    const unsigned yrel = (unsigned)((((s_row / BTH) * H + s_row % BTH) * W + 8 * s_wg) * y_ld + 8 * s_half);
    const unsigned hrel = (unsigned)((((hy_row / BTH) * H + hy_row % BTH) * W + (h_side ? BTW + 1 : 0)) * y_ld + 8 * h_half);
    const bool ch_full = xc0 + 16 <= xcn && cob * 16 + 16 <= Co;
    auto load_tile = [&](int tile) {
        const int w0 = (tile % tilesW) * BTW;
        tile /= tilesW;
        const int d0 = (tile % tilesD) * BTD;
        tile /= tilesD;
        const int h0 = (tile % tilesH) * BTH;
        const int n = tile / tilesH;
        if (d0 >= 1 && d0 + BTD < D && h0 >= 1 && h0 + BTH < H && w0 >= 1 && w0 + BTW < W && ch_full) {
            // interior tile (wave-uniform): scalar bases + precomputed lane offsets, no coordinates, no masks
            const bf16_t* xb = x + ((((int64_t)n * D + d0 - 1) * H + h0 - 1) * W + w0) * x_ld + xc0;
            const bf16_t* yb = dy + ((((int64_t)n * D + d0) * H + h0) * W + w0) * y_ld + cob * 16;
#pragma unroll
            for (int j = 0; j < 8; ++j) vx[j] = ldg4u(xb + j * x_ld + xrel);
            if (is_y) {
#pragma unroll
                for (int j = 0; j < 8; ++j) vy[j] = ldg4u(yb + j * y_ld + yrel);
            }
            if (is_h) vh = ldg4u(yb - y_ld + hrel);
            return;
        }
        {   // ---- X: 8 voxels x 8 channels per lane
            const int gd = d0 - 1 + s_row / BHH, gh = h0 - 1 + s_row % BHH;
            const int c0 = xc0 + 8 * s_half;
            const bool rok = (unsigned)gd < (unsigned)D && (unsigned)gh < (unsigned)H && c0 < xcn;
            const bf16_t* src = x + ((((int64_t)n * D + (rok ? gd : 0)) * H + (rok ? gh : 0)) * W) * x_ld + (c0 < xcn ? c0 : 0);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int gw = w0 + 8 * s_wg + j;
                // unconditional, explicitly global; an out-of-volume piece reads g_zero16 (address select: no wait on the load here)
                vx[j] = ldg4u((rok && gw < W) ? src + (int64_t)gw * x_ld : reinterpret_cast<const bf16_t*>(g_zero16));
            }
        }
        if (is_y) {   // ---- dY: 12 rows x 4 w-groups x 2 channel halves
            const int gd = d0 + s_row / BTH, gh = h0 + s_row % BTH;
            const int c0 = cob * 16 + 8 * s_half;
            const bool rok = gd < D && gh < H && c0 < Co;       // host guarantees Co % 8 == 0
            const bf16_t* src = dy + ((((int64_t)n * D + (rok ? gd : 0)) * H + (rok ? gh : 0)) * W) * y_ld + (c0 < Co ? c0 : 0);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int gw = w0 + 8 * s_wg + j;
                vy[j] = ldg4u((rok && gw < W) ? src + (int64_t)gw * y_ld : reinterpret_cast<const bf16_t*>(g_zero16));
            }
        }
        if (is_h) {   // ---- dY W-halo voxels w0 - 1 and w0 + 32
            const int gd = d0 + hy_row / BTH, gh = h0 + hy_row % BTH, gw = h_side ? w0 + BTW : w0 - 1;
            const int c0 = cob * 16 + 8 * h_half;
            const bool ok = gd < D && gh < H && (unsigned)gw < (unsigned)W && c0 < Co;
            vh = ldg4u(ok ? dy + ((((int64_t)n * D + gd) * H + gh) * W + gw) * y_ld + c0 : reinterpret_cast<const bf16_t*>(g_zero16));
        }
    };
    auto store_tile = [&]() {
        uint4 o[8];
        transpose8x8_bf16(vx, o);
        char* dline = xs + (s_row * 16 + 8 * s_half) * DLS + 16 * rot_slot(s_wg, 8 * s_half);   // 8 channels share a rotation
#pragma unroll
        for (int c = 0; c < 8; ++c) *reinterpret_cast<uint4*>(dline + c * DLS) = o[c];
        if (is_y) {
            transpose8x8_bf16(vy, o);
            char* yline = ys + (s_row * 16 + 8 * s_half) * DLS + 16 * rot_slot(s_wg, 8 * s_half);
#pragma unroll
            for (int c = 0; c < 8; ++c) *reinterpret_cast<uint4*>(yline + c * DLS) = o[c];
        } else if (is_h) {   // w0 - 1: high half of dword 0;  w0 + 32: low half of dword 1
            const unsigned* hw = reinterpret_cast<const unsigned*>(&vh);
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const unsigned short val = (unsigned short)((c & 1) ? (hw[c >> 1] >> 16) : (hw[c >> 1] & 0xffffu));
                *reinterpret_cast<unsigned short*>(yh + (hy_row * 16 + 8 * h_half + c) * 8 + (h_side ? 4 : 2)) = val;
            }
        }
    };

    const TileWalk tw = tile_walk(ntiles);
    if (tw.count > 0) load_tile(tw.first);
    for (int k = 0; k < tw.count; ++k) {
        __syncthreads();   // the previous tile's MFMAs are done with the LDS tile
        store_tile();
        __syncthreads();
        if (k + 1 < tw.count) load_tile(tw.first + (k + 1) * tw.stride);
        __builtin_amdgcn_sched_barrier(0);

        // ---- 3 output rows per wave x 9 (kd, kh) x 3 kw MFMAs
        // the wave's three rows are consecutive in h: constant-stride row pointers, every tap an immediate offset
        const char* xrow = xrow0;
        const char* yrow = yrow0;
        const char* ypl = ypl0;
        const char* ynr = ynr0;
#pragma unroll 1
        for (int r = 0; r < BYR / 4; ++r, xrow += 16 * DLS, yrow += 16 * DLS, ypl += pl_step, ynr += nr_step) {
            const uint4 b1 = *reinterpret_cast<const uint4*>(yrow);            // dY[u], the lane's eight voxels
            const unsigned pl = *reinterpret_cast<const unsigned*>(ypl);        // high half = dY[first - 1]
            const unsigned nr = *reinterpret_cast<const unsigned*>(ynr);        // low half = dY[last + 1]
            uint4 bm, bp;   // dY[u - 1], dY[u + 1]
            bm.x = __builtin_amdgcn_alignbyte(b1.x, pl, 2);
            bm.y = __builtin_amdgcn_alignbyte(b1.y, b1.x, 2);
            bm.z = __builtin_amdgcn_alignbyte(b1.z, b1.y, 2);
            bm.w = __builtin_amdgcn_alignbyte(b1.w, b1.z, 2);
            bp.x = bm.y;
            bp.y = bm.z;
            bp.z = bm.w;
            bp.w = __builtin_amdgcn_alignbyte(nr, b1.w, 2);
            const bf16x8_t b0v = __builtin_bit_cast(bf16x8_t, bp), b1v = __builtin_bit_cast(bf16x8_t, b1),
                           b2v = __builtin_bit_cast(bf16x8_t, bm);
            if (BIAS) acc[TG] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, b1v, acc[TG], 0, 0, 0);
#pragma unroll
            for (int kdh = 0; kdh < 9; ++kdh) {
                const int lrow = ((kdh / 3) * BHH + kdh % 3) * 16;   // compile-time after unrolling
                const bf16x8_t g = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(xrow + lrow * DLS));
                acc[kdh * 3 + 0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(g, b0v, acc[kdh * 3 + 0], 0, 0, 0);   // X[u] dY[u+1]
                acc[kdh * 3 + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(g, b1v, acc[kdh * 3 + 1], 0, 0, 0);   // X[u] dY[u]
                acc[kdh * 3 + 2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(g, b2v, acc[kdh * 3 + 2], 0, 0, 0);   // X[u] dY[u-1]
            }
        }
    }

    // combine the 4 waves in a fixed order through LDS, then one partial per workgroup
    __syncthreads();
    float* red = lds;  // [TGA][256]
    for (int w = 0; w < 4; ++w) {
        if (wv == w) {
#pragma unroll
            for (int t = 0; t < TGA; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int o = t * 256 + (4 * kq + r) * 16 + li;
                    red[o] = (w == 0) ? acc[t][r] : red[o] + acc[t][r];
                }
        }
        __syncthreads();
    }
    float* out = part + (((size_t)blockIdx.x * gridDim.y + cit) * gridDim.z + cob) * (TGA * 256);
    for (int i = tid; i < TGA * 256; i += 256) out[i] = red[i];
}

// ------------------------------------------------------------------ bf16 weight gradient, marching along d, staged by LDS-DMA
// The tile kernel above re-reads its input 2.7x (a 2 x 6-row tile needs 4 x 8 rows of X) and is bound by what a CU can pull
// through its memory pipeline (MFMA busy 29 %).  Here a workgroup owns a COLUMN of the volume — 8 rows x 32 voxels
// — and marches through a segment of d planes with the last planes of X in an LDS ring: per output plane it fetches ONE new
// plane of X (10 rows) and one of dY (8 rows + the W halo).  Round 2 staged the column through registers (eight 16-byte loads
// with selected addresses, 32 v_perm of an 8x8 transpose and eight ds_write_b128 per staging lane and plane): ~100 vector
// instructions per plane which, beside the other workgroup's MFMAs, issue only each 14-28 cycles — 4 900 cycles
// per plane step for 1 800 cycles of MFMA work per SIMD (MFMA busy 0.30, 0.59 ms on 48 -> 16 at 2 x 160x192x160).
// Here NO vector instruction touches the data on its way in: the rows are copied as they lie in memory (channels last: one
// voxel = 32 bytes of a 16-channel tile) by LDS-DMA, and the transposition the MFMA needs — a lane's operand is a run of voxels
// of ONE channel — is done by the read: ds_read_b64_tr_b16 hands lane i of a 16-lane group channel i of four consecutive voxels
// (cdna_hip_programming.md T10).  With the address  row + 512*j + 8*lane  (j = 0, 1) a wave reads 512 contiguous bytes per
// instruction (conflict-free), and k-group kq of the K = 32 operand holds voxels 4kq..4kq+3 and 16+4kq..16+4kq+3 of the row —
// the same order in X and dY, so the sum over k is the sum over the row's 32 voxels.  The kw taps:
//   dW[kd,kh,kw] = sum_v X[v + kw - 1] dY[v]  =  sum_u X[u] dY[u + 1 - kw]
// i.e. dY read one voxel (32 bytes) to the left / right: an address, not a shuffle; the sums are re-partitioned between
// W-neighbouring columns — a column takes the products of ITS X voxels, with dY[w0-1] and dY[w0+32] read from the neighbours
// (zero outside the volume) — so X has no W halo and dY a one-voxel one.  Tasks = (sample, d-segment, column); accumulators
// (27 taps x (16 ci x 16 co) per wave, +1 for dbias fed with A = 1) and partial layout as in the tile kernel.
//   LDS: X planes (10 rows x 1 KiB) and dY planes (8 rows x 34 voxels, 1 088 B per row) in rings of four: the plane being
//   read + three in flight.  A wave keeps the X fragments of planes t-1 and t in REGISTERS from the steps that read them, so
//   LDS holds one live plane of each tensor instead of three + one: the same 78 KB carry three planes of lead instead of two —
//   on one-ci-tile layers the kernel's rate is (bytes in flight) / (loaded latency, ~3.7 us), DESIGN.md §4.4.
//   step t:  wait for the wave's pieces of step t-3 (counted vmcnt: those of steps t-2 and t-1 may still fly), barrier,
//            issue X plane t+4 and dY plane t+3 (19 pieces of 1 KiB per workgroup, 5 or 4 per wave, per-lane offsets constant
//            for the column; a plane outside the segment's range is a resource of zero records: zeros),
//            read the fragments of X plane t+1, multiply plane t (X planes t-1, t from registers, t+1; dY plane t).
// Measured against the register-staged kernel (tools/r03_wgt.sh, one box): 48 -> 16 0.590 -> 0.503 ms, 16 -> 16 0.232 -> 0.195,
// 8 -> 16 0.215 -> 0.180, 96 -> 32 at 80x96x80 0.358 -> 0.302, 16 -> 16 at 512 x 32^3 0.333 -> 0.287 with rings of five / three
// planes and two planes of lead; with the X fragments of two planes in registers and three planes of lead 0.457 / 0.186 / 0.168 /
// 0.291 / 0.26 ms.  What binds it is the
// fabric: with the MFMAs compiled out the 48 -> 16 layer still takes 0.415 ms (its three ci-tile workgroups each fetch dY: PMC
// 2.0x the algorithmic bytes, 5 TB/s), with the DMA compiled out 0.280 ms; with every workgroup on one L2-resident column the
// DMA alone runs at 14 TB/s.  (Workgroups of one task's ci-tiles share an XCD — tools/microbench/xcc_probe.hip — yet run in lock
// step and miss together; a start skew does not survive, the follower catches up.  One workgroup per CU taking all three ci-tiles
// against a single staging of dY — scatter form, dY in a five-plane ring, X triple-buffered, 138 KB of LDS — was built and is
// parity-green but slower: 0.72 ms with six waves of four rows (one or two waves per SIMD expose every fragment read and the
// step barrier), and twelve waves of two rows do not fit 170 registers: hipcc spills inside the MFMA loop.)
constexpr int MTH = 8, MXR = MTH + 2;                 // output rows / X rows per plane
constexpr int kMarchSeg = 40;                         // planes per task at most (5 fill steps per task); shorter for small volumes
constexpr int TXROW = BTW * 32;                       //  1 024 B  one X row: 32 voxels x 16 channels
constexpr int TXP = MXR * TXROW;                      // 10 240 B  one X plane
constexpr int TXSLOTS = 4;                            //           ring: the plane being read + three in flight
constexpr int TYROW = (BTW + 2) * 32;                 //  1 088 B  one dY row with its two W-halo voxels
constexpr int TYPIECES = (MTH * TYROW + 1023) / 1024; //  9 pieces of 1 KiB (the ninth: lanes 0..31)
constexpr int TYP = TYPIECES * 1024;                  //  9 216 B  one dY plane (8 704 used)
constexpr int TYSLOTS = 4;
constexpr int TLDS = TXSLOTS * TXP + TYSLOTS * TYP;   // 77 824 B: two workgroups per CU
constexpr int TNPC = (MXR + TYPIECES + 3) / 4;        // DMA pieces per wave and step at most (5)

typedef short s16x4_t __attribute__((ext_vector_type(4)));
// the K = 32 operand of the lane from a raw [voxel][16 channels] row: `a` = LDS byte address of the row + 8 * lane
__device__ __forceinline__ bf16x8_t tr_frag(unsigned a) {
    typedef __attribute__((address_space(3))) s16x4_t* lp;
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(a));
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(a + 512u));
    return __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

template <bool BIAS>
__global__ void __launch_bounds__(256, 2)
conv_mfma_wgrad_bf16t_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy, float* __restrict__ part, int N,
                             int D, int H, int W, int Ci, int x_ld, int Co, int y_ld, int nseg, int tilesH, int tilesW,
                             int ntasks, const bf16_t* __restrict__ x2, int x2_ld, int ksplit, int segl) {
    constexpr int TG = 27, TGA = TG + (BIAS ? 1 : 0);
    int xc0 = (int)blockIdx.y * 16, xcn = Ci;   // conv over cat((x, x2)): the ci-tile lives in ONE of the two tensors
    if (x2 != nullptr) {
        if (xc0 >= ksplit) { x = x2; x_ld = x2_ld; xc0 -= ksplit; xcn = Ci - ksplit; }
        else xcn = ksplit;
    }
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const unsigned xs0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)lds), ys0 = xs0 + TXSLOTS * TXP;
    const int cit = blockIdx.y, cob = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kq = lane >> 4;

    f32x4 acc[TGA];
#pragma unroll
    for (int t = 0; t < TGA; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8_t ones;
#pragma unroll
    for (int i = 0; i < 8; ++i) ones[i] = (bf16_t)1.0f;

    // operand addresses of the wave's two output rows (2 wv, 2 wv + 1): X rows 2 wv .. 2 wv + 3 of a plane, dY voxel u at + 32 (u + 1)
    const unsigned xfrag = xs0 + (unsigned)(2 * wv * TXROW + 8 * lane);        // + slot * TXP + row * TXROW
    const unsigned yfrag = ys0 + (unsigned)(2 * wv * TYROW + 32 + 8 * lane);   // + slot * TYP + row * TYROW + 32 * (1 - kw)

    const TileWalk tw = tile_walk(ntasks);
    for (int k = 0; k < tw.count; ++k) {
        int task = tw.first + k * tw.stride;
        const int w0 = (task % tilesW) * BTW;
        task /= tilesW;
        const int h0 = (task % tilesH) * MTH;
        task /= tilesH;
        const int seg = task % nseg, n = task / nseg;
        const int dA = seg * segl, dB = min(D, dA + segl);
        const int xlo = max(dA - 1, 0), xhi = min(dB, D - 1);   // X planes the segment consumes

        // the wave's pieces of a plane step: ids wv, wv + 4, ... < 19; id < 10 = X row id, else piece id - 10 of the dY plane.
        // Byte offset of the lane's 16 bytes from the plane's origin voxel, the out-of-volume value folded in: constant for the column.
        unsigned vof[TNPC];
#pragma unroll
        for (int i = 0; i < TNPC; ++i) {
            const int id = wv + 4 * i;
            vof[i] = kDmaOob;
            if (id < MXR) {   // wave-uniform
                const int vox = lane >> 1, half = lane & 1;
                const int gh = h0 - 1 + id, gw = w0 + vox;
                if ((unsigned)gh < (unsigned)H && gw < W && xc0 + 8 * half < xcn) vof[i] = (unsigned)(((id * W + vox) * x_ld + 8 * half) * 2);
            } else if (id < MXR + TYPIECES) {
                const int f = (id - MXR) * 64 + lane, row = f / (2 * (BTW + 2)), pc = f - row * (2 * (BTW + 2));
                const int vox = pc >> 1, half = pc & 1;
                const int gh = h0 + row, gw = w0 - 1 + vox;
                if (row < MTH && gh < H && (unsigned)gw < (unsigned)W && cob * 16 + 8 * half < Co)
                    vof[i] = (unsigned)(((row * W + vox) * y_ld + 8 * half) * 2);
            }
        }
        // plane origins: X at (h0 - 1, w0), dY at (h0, w0 - 1); a lane whose voxel lies outside the volume never dereferences them
        const int64_t xplane = (int64_t)H * W * x_ld * 2, yplane = (int64_t)H * W * y_ld * 2;
        const unsigned long long xorg = (unsigned long long)(x + ((((int64_t)n * D) * H + (h0 - 1)) * W + w0) * x_ld + xc0);
        const unsigned long long yorg = (unsigned long long)(dy + ((((int64_t)n * D) * H + h0) * W + (w0 - 1)) * y_ld + cob * 16);

        auto rsrc = [&](unsigned long long org, bool ok) {
            i32x4 rs;
            rs[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)(org & 0xffffffffu));
            rs[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)((org >> 32) & 0xffffu));
            rs[2] = ok ? (int)kDmaRecords : 0;
            rs[3] = 0x00020000;
            return rs;
        };
        // X fragments of planes t-1 and t stay in registers from the steps that read them: the LDS keeps ONE live plane of X and
        // of dY and three in flight of each (lead 3 instead of 2 in the same 78 KB)
        const int t0 = dA - 5;
        int sx = ((t0 + 4) % TXSLOTS + TXSLOTS) % TXSLOTS;   // ring slot of X plane t + 4
        int sy = ((t0 + 3) % TYSLOTS + TYSLOTS) % TYSLOTS;   // ring slot of dY plane t + 3
        bf16x8_t xfA[4], xfB[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) xfA[i] = xfB[i] = ones;
        for (int t = t0; t < dB; ++t) {
            if (wv == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            {
                const int qx = t + 4, qy = t + 3;
                const i32x4 rsx = rsrc(xorg + (unsigned long long)((int64_t)qx * xplane), qx >= xlo && qx <= xhi);
                const i32x4 rsy = rsrc(yorg + (unsigned long long)((int64_t)qy * yplane), qy >= dA && qy < dB);
                const unsigned xdst = xs0 + (unsigned)(sx * TXP), ydst = ys0 + (unsigned)(sy * TYP);
#pragma unroll
                for (int i = 0; i < TNPC; ++i) {
                    const int id = wv + 4 * i;
                    if (id < MXR) lds_dma16(vof[i], rsx, xdst + (unsigned)(id * 1024));
                    else if (id < MXR + TYPIECES - 1) lds_dma16(vof[i], rsy, ydst + (unsigned)((id - MXR) * 1024));
                    else if (id == MXR + TYPIECES - 1) {
                        if (lane < (MTH * TYROW - (TYPIECES - 1) * 1024) / 16) lds_dma16(vof[i], rsy, ydst + (unsigned)((id - MXR) * 1024));
                    }
                }
                sx = sx + 1 == TXSLOTS ? 0 : sx + 1;   // now the slot of plane t+5 = the slot of plane t+1
                sy = sy + 1 == TYSLOTS ? 0 : sy + 1;   // now the slot of plane t+4 = the slot of plane t
            }
            if (t >= dA - 2) {
                bf16x8_t xfC[4];
                const unsigned xb = xfrag + (unsigned)(sx * TXP);
#pragma unroll
                for (int i = 0; i < 4; ++i) xfC[i] = tr_frag(xb + (unsigned)(i * TXROW));
                if (t >= dA) {
                    const unsigned yb = yfrag + (unsigned)(sy * TYP);
                    bf16x8_t dyf[2][3];
#pragma unroll
                    for (int r = 0; r < 2; ++r)
#pragma unroll
                        for (int kw = 0; kw < 3; ++kw) dyf[r][kw] = tr_frag(yb + (unsigned)(r * TYROW + 32 * (1 - kw)));   // dY[u + 1 - kw]
                    if constexpr (BIAS) {
                        acc[TG] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, dyf[0][1], acc[TG], 0, 0, 0);
                        acc[TG] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, dyf[1][1], acc[TG], 0, 0, 0);
                    }
#pragma unroll
                    for (int kd = 0; kd < 3; ++kd)
#pragma unroll
                        for (int r = 0; r < 2; ++r)
#pragma unroll
                            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                                for (int kw = 0; kw < 3; ++kw) {
                                    const bf16x8_t a = kd == 0 ? xfA[r + kh] : kd == 1 ? xfB[r + kh] : xfC[r + kh];
                                    acc[(kd * 3 + kh) * 3 + kw] =
                                        __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, dyf[r][kw], acc[(kd * 3 + kh) * 3 + kw], 0, 0, 0);
                                }
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) { xfA[i] = xfB[i]; xfB[i] = xfC[i]; }
            }
        }
    }
    // every DMA piece has landed (the last steps' zero planes too), every wave is done reading: the ring becomes the reduction buffer
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    float* red = lds;  // [TGA][256]
    for (int w = 0; w < 4; ++w) {
        if (wv == w) {
#pragma unroll
            for (int t = 0; t < TGA; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int o = t * 256 + (4 * kq + r) * 16 + li;
                    red[o] = (w == 0) ? acc[t][r] : red[o] + acc[t][r];
                }
        }
        __syncthreads();
    }
    float* out = part + (((size_t)blockIdx.x * gridDim.y + cit) * gridDim.z + cob) * (TGA * 256);
    for (int i = tid; i < TGA * 256; i += 256) out[i] = red[i];
}

// ------------------------------------------------------------------ weight gradient, version 6 (fp32, Cin % 16 == 0)
// The transposed tile for fp32: a line holds 16 voxels, and the MFMA K order is permuted so that k-group kq of the four k-steps
// of a row owns voxels 4kq..4kq+3 — ONE ds_read_b128 feeds four v_mfma_f32_16x16x4_f32 k-steps.  The kw = 0 / 2 taps take the
// same X fragment against dY shifted by one voxel: pure register selection from (left neighbour, fragment, right neighbour) of
// dY, done once per row.  A 2x6x16-voxel tile costs a wave 3 rows x (3 + 9) = 36 LDS reads for 324 MFMAs (v4 issued one 4-byte
// LDS read per MFMA, round 1's v6 84 reads); every non-MFMA instruction shows up as idle MFMA time.  Staging
// transposes 4 voxels x 4 channels per lane by register renaming.  Same accumulators and partial layout as v4.
constexpr int FTW = 16;   // voxels per line

// 16-byte load through an explicitly GLOBAL pointer (global_load_dwordx4: vmcnt only, never lgkmcnt)
__device__ __forceinline__ float4 ldg4(const float* p) {
    typedef float gf32x4 __attribute__((ext_vector_type(4)));
    const gf32x4 v = *(const __attribute__((address_space(1))) gf32x4*)p;
    return make_float4(v[0], v[1], v[2], v[3]);
}

//
// Eight-channel operands (Modified3DUNet's first level, modified_3dunet.py:33-55: 8 -> 8 twice, 16 -> 8) would leave half of a
// 16-row / 16-column operand empty.  CO8 (Co == 8): columns 8..15 of the dY operand carry the SAME eight channels for the next kw
// tap — lane li reads channel li & 7 and selects dY[u+1] (kw 0, columns 0..7) or dY[u] (kw 1, columns 8..15); a second operand
// carries kw 2: two MFMAs per (kd, kh) and k-step instead of three.  CI8 (Ci == 8): rows 8..15 of the X operand carry the same
// eight channels for the next (kd, kh) row pair — (kd,kh) = 2p + (li >> 3), five fragments instead of nine.  8 -> 8 then needs
// 10 MFMAs per k-step instead of the 27 of a half-empty 16 x 16 tile (and of the 14 of the older v3 kernel); the unused halves of
// the last pair / the kw-2 operand are duplicates that wgrad_mfma_reduce_kernel drops (accumulator -> tap map: wg6_tap()).
__host__ __device__ constexpr int wg6_groups(bool ci8, bool co8) { return (ci8 ? 5 : 9) * (co8 ? 2 : 3); }
// tap of element (row, col) of accumulator tg, or -1 (duplicate / padding)
__host__ __device__ inline int wg6_tap(bool ci8, bool co8, int tg, int row, int col) {
    const int nb = co8 ? 2 : 3, pa = tg / nb, q = tg % nb;
    int kdh = pa, kw = q;
    if (ci8) {
        kdh = 2 * pa + (row >> 3);
        if (kdh > 8) return -1;
    }
    if (co8) {
        kw = q == 0 ? (col >> 3) : 2;
        if (q == 1 && (col >> 3)) return -1;
    }
    return kdh * 3 + kw;
}

template <bool BIAS, bool CI8 = false, bool CO8 = false>
__global__ void __launch_bounds__(256, 2)
conv_mfma_wgrad6_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ part, int N, int D,
                        int H, int W, int Ci, int x_ld, int Co, int y_ld, int tilesD, int tilesH, int tilesW, int ntiles,
                        const float* __restrict__ x2, int x2_ld, int ksplit) {
    constexpr int NA = CI8 ? 5 : 9, NB = CO8 ? 2 : 3;
    constexpr int TG = NA * NB, TGA = TG + (BIAS ? 1 : 0);
    // conv over cat((x, x2)): this workgroup's 16-channel ci-tile lives in ONE of the two tensors — rebind x / pitch / channel origin
    int xc0 = (int)blockIdx.y * 16;
    if (x2 != nullptr && xc0 >= ksplit) { x = x2; x_ld = x2_ld; xc0 -= ksplit; }
    extern __shared__ __attribute__((aligned(16))) float lds[];
    char* xs = reinterpret_cast<char*>(lds);
    char* ys = xs + BXS;
    char* yh = ys + BYS;

    const int cit = blockIdx.y, cob = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int li = lane & 15, kq = lane >> 4;

    f32x4 acc[TGA];
#pragma unroll
    for (int t = 0; t < TGA; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

    // staging roles: a unit = 4 consecutive voxels x 4 channels; X has 32 rows x 4 w-groups x 4 quads = 512 units (2 per
    // lane), dY 12 rows x 4 x 4 = 192 units (lanes < 192), its W halo 12 rows x 2 sides x 4 quads = 96 voxels (lanes < 96)
    const int s_q = tid & 3, s_wg = (tid >> 2) & 3, s_row = tid >> 4;      // unit u = tid (+256): row = s_row (+16)
    const int h_q = tid & 3, h_side = (tid >> 2) & 1, h_row = tid >> 3;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const int orow0 = wv * (BYR / 4);   // first of the wave's three output rows (same d-plane, consecutive h)
    const int ar = CI8 ? (li & 7) : li, ahs = CI8 ? (li >> 3) : 0;   // X operand row li: channel, (kd, kh) half
    const int bc = CO8 ? (li & 7) : li, bhs = CO8 ? (li >> 3) : 0;   // dY operand column li: channel, kw half
    const char* const xrow0 = xs + (((orow0 / BTH) * BHH + orow0 % BTH) * 16 + ar) * DLS + 16 * rot_slot(kq, ar);   // tap (kd,kh) = (0,0)
    int aoff[NA];   // CI8: the lane's line offset of pair p, (kd, kh) = 2p + ahs (the ninth has no partner: a duplicate)
#pragma unroll
    for (int pa = 0; pa < NA; ++pa) {
        const int kdh = CI8 ? (2 * pa + ahs < 9 ? 2 * pa + ahs : 8) : pa;
        aoff[pa] = ((kdh / 3) * BHH + kdh % 3) * 16 * DLS;
    }
    const int yline0 = orow0 * 16 + bc;
    const char* const yrow0 = ys + yline0 * DLS + 16 * rot_slot(kq, bc);
    // voxel before / after the lane's four: last float of the previous / first float of the next k-group's slot, or the W halo
    const char* const ypl0 = kq == 0 ? yh + yline0 * 8 : ys + yline0 * DLS + 16 * rot_slot(kq - 1, bc) + 12;
    const char* const ynr0 = kq == 3 ? yh + yline0 * 8 + 4 : ys + yline0 * DLS + 16 * rot_slot(kq + 1, bc);
    const int pl_step = kq == 0 ? 16 * 8 : 16 * DLS, nr_step = kq == 3 ? 16 * 8 : 16 * DLS;

    // next tile's pieces: fetched into registers while the current tile is multiplied (HBM/L2 latency hidden), written to
    // the single LDS tile between two barriers after it
    float4 vx[2][4], vh, vy[4];
    // Per-lane element offsets of its pieces from the tile's origin voxels, computed once: a piece's address is then a
    // wave-uniform tile base + a 32-bit lane offset, with no per-tile vector multiplies or 64-bit mads.
    // X origin = voxel (d0-1, h0-1, w0), dY origin = (d0, h0, w0).  Lanes without a piece (no such channel quad / row) read a
    // duplicate of an existing one and do not store it: EVERY lane issues the same 13 loads, unconditionally.
    // Two forms of the staging.  The 16-channel kernel keeps round 1's (load_tile_base): an interior tile is a scalar base plus
    // the lane offsets, wait-free; only border tiles (29 % at 160x192x160) select on loaded values, which makes hipcc wait for the
    // loads on the spot.  The eight-channel variants (load_tile_8) issue the same 13 loads in every lane, unconditionally —
    // lanes without a piece read a duplicate — and never look at a loaded value: out-of-volume pieces read a safe in-volume voxel,
    // their validity goes into a bit mask, and store_tile_8() zeroes them after the MFMAs (a load under a divergent `if` made
    // hipcc wait inside the interior path of these variants).  The same form for the 16-channel kernel measured 8 % SLOWER on the
    // 48 -> 16 layer (3.35 -> 3.63 ms): its per-lane 64-bit address arithmetic does not hide behind fp32 MFMAs.
    unsigned xrel[2], yrel;
    const int xq = CI8 ? (s_q & 1) : s_q, yq = CO8 ? (s_q & 1) : s_q, hq = CO8 ? (h_q & 1) : h_q;   // an existing channel quad
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int row = s_row + 16 * u;
        xrel[u] = (unsigned)((((row / BHH) * H + row % BHH) * W + 4 * s_wg) * x_ld + 4 * xq);
    }
    const unsigned xsafe = (unsigned)(((H + 1) * W) * x_ld);   // the tile's first output voxel (always in the volume)
    const bool yrow_ok = tid < BYR * 16, hrow_ok = tid < BYR * 8;
    yrel = yrow_ok ? (unsigned)((((s_row / BTH) * H + s_row % BTH) * W + 4 * s_wg) * y_ld + 4 * yq) : 0u;
    // halo voxel relative to (d0, h0, w0 - 1): never negative
    const unsigned hrel = hrow_ok ? (unsigned)((((h_row / BTH) * H + h_row % BTH) * W + (h_side ? FTW + 1 : 0)) * y_ld + 4 * hq) : (unsigned)y_ld;
    const bool co_full = CO8 || cob * 16 + 16 <= Co;
    const bool xon = !CI8 || s_q < 2, yon = yrow_ok && (!CO8 || s_q < 2), hon = hrow_ok && (!CO8 || h_q < 2);   // lanes that store
    unsigned okbits = ~0u;   // of the tile in the registers: bit 4u+j X piece (u, j), bit 8+j dY piece j, bit 12 the halo voxel
    bool border = false;     // ... and whether any lane has a zero bit (wave-uniform)
    auto load_tile_base = [&](int tile) {   // both operands 16 channels wide: the round-1 form, at the register limit as it is
        const int w0 = (tile % tilesW) * FTW;
        tile /= tilesW;
        const int d0 = (tile % tilesD) * BTD;
        tile /= tilesD;
        const int h0 = (tile % tilesH) * BTH;
        const int n = tile / tilesH;
        // wave-uniform bases; the X base may point before the tensor (d0 = 0 ...) and is only dereferenced at valid offsets
        const float* xb = x + ((((int64_t)n * D + d0 - 1) * H + h0 - 1) * W + w0) * x_ld + xc0;
        const int c0 = cob * 16 + 4 * s_q;
        const float* yb = dy + ((((int64_t)n * D + d0) * H + h0) * W + w0) * y_ld + cob * 16;
        // Interior tile (the common case; wave-uniform test on scalars): every piece is in the volume, so a piece's address is
        // a scalar base (tile origin + j voxels) plus the lane's precomputed 32-bit offset — no per-lane coordinates, no masks.
        // The general path below costs ~300 instructions per tile against the tile's 336 MFMAs per wave, and none of them hides
        // behind the fp32 MFMA.
        if (d0 >= 1 && d0 + BTD < D && h0 >= 1 && h0 + BTH < H && w0 >= 1 && w0 + FTW < W && co_full) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float* xj = xb + j * x_ld;   // scalar
                vx[0][j] = ldg4(xj + xrel[0]);
                vx[1][j] = ldg4(xj + xrel[1]);
            }
            if (tid < BYR * 16) {
#pragma unroll
                for (int j = 0; j < 4; ++j) vy[j] = ldg4(yb + j * y_ld + yrel);
            }
            if (tid < BYR * 8) vh = ldg4(yb - y_ld + hrel);
            return;
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {   // ---- X
            const int row = s_row + 16 * u;
            const int gd = d0 - 1 + row / BHH, gh = h0 - 1 + row % BHH;
            const bool rok = (unsigned)gd < (unsigned)D && (unsigned)gh < (unsigned)H;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool ok = rok && w0 + 4 * s_wg + j < W;
                // (the address-select form of the bf16 kernel — out-of-volume pieces read g_zero16, no wait here — costs this kernel
                // two more registers than it has: 2 spills, 48 -> 16 layer 3.33 -> 3.47 ms)
                const float4 t = ldg4(xb + (ok ? xrel[u] + (unsigned)(j * x_ld) : xsafe));
                vx[u][j] = ok ? t : zero4;
            }
        }
        if (tid < BYR * 16) {   // ---- dY
            const int gd = d0 + s_row / BTH, gh = h0 + s_row % BTH;
            const bool rok = gd < D && gh < H && c0 < Co;       // host guarantees Co % 4 == 0
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool ok = rok && w0 + 4 * s_wg + j < W;
                const float4 t = ldg4(yb + (ok ? yrel + (unsigned)(j * y_ld) : 0u));
                vy[j] = ok ? t : zero4;
            }
        }
        if (tid < BYR * 8) {   // ---- dY W-halo voxels w0 - 1 / w0 + 16
            const int gd = d0 + h_row / BTH, gh = h0 + h_row % BTH, gw = h_side ? w0 + FTW : w0 - 1;
            const int hc = cob * 16 + 4 * h_q;
            const bool ok = gd < D && gh < H && (unsigned)gw < (unsigned)W && hc < Co;
            const float4 t = ldg4(dy + ((((int64_t)n * D + (ok ? gd : d0)) * H + (ok ? gh : h0)) * W + (ok ? gw : w0)) * y_ld + (ok ? hc : 0));
            vh = ok ? t : zero4;
        }
    };
    auto store_tile_base = [&]() {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            char* dst = xs + ((s_row + 16 * u) * 16 + 4 * s_q) * DLS + 16 * rot_slot(s_wg, 4 * s_q);   // 4 channels share a rotation
            *reinterpret_cast<float4*>(dst) = make_float4(vx[u][0].x, vx[u][1].x, vx[u][2].x, vx[u][3].x);
            *reinterpret_cast<float4*>(dst + DLS) = make_float4(vx[u][0].y, vx[u][1].y, vx[u][2].y, vx[u][3].y);
            *reinterpret_cast<float4*>(dst + 2 * DLS) = make_float4(vx[u][0].z, vx[u][1].z, vx[u][2].z, vx[u][3].z);
            *reinterpret_cast<float4*>(dst + 3 * DLS) = make_float4(vx[u][0].w, vx[u][1].w, vx[u][2].w, vx[u][3].w);
        }
        if (tid < BYR * 16) {
            char* dst = ys + (s_row * 16 + 4 * s_q) * DLS + 16 * rot_slot(s_wg, 4 * s_q);
            *reinterpret_cast<float4*>(dst) = make_float4(vy[0].x, vy[1].x, vy[2].x, vy[3].x);
            *reinterpret_cast<float4*>(dst + DLS) = make_float4(vy[0].y, vy[1].y, vy[2].y, vy[3].y);
            *reinterpret_cast<float4*>(dst + 2 * DLS) = make_float4(vy[0].z, vy[1].z, vy[2].z, vy[3].z);
            *reinterpret_cast<float4*>(dst + 3 * DLS) = make_float4(vy[0].w, vy[1].w, vy[2].w, vy[3].w);
        }
        if (tid < BYR * 8) {
            char* dst = yh + (h_row * 16 + 4 * h_q) * 8 + 4 * h_side;
            *reinterpret_cast<float*>(dst) = vh.x;
            *reinterpret_cast<float*>(dst + 8) = vh.y;
            *reinterpret_cast<float*>(dst + 16) = vh.z;
            *reinterpret_cast<float*>(dst + 24) = vh.w;
        }
    };

    auto load_tile_8 = [&](int tile) {
        const int w0 = (tile % tilesW) * FTW;
        tile /= tilesW;
        const int d0 = (tile % tilesD) * BTD;
        tile /= tilesD;
        const int h0 = (tile % tilesH) * BTH;
        const int n = tile / tilesH;
        // wave-uniform bases; the X base may point before the tensor (d0 = 0 ...) and is only dereferenced at valid offsets
        const float* xb = x + ((((int64_t)n * D + d0 - 1) * H + h0 - 1) * W + w0) * x_ld + xc0;
        const float* yb = dy + ((((int64_t)n * D + d0) * H + h0) * W + w0) * y_ld + cob * 16;
        unsigned xo[2][4], yo[4], ho = hrel, bits = ~0u;
        // Interior tile (the common case; wave-uniform test on scalars): every piece is in the volume.  The general path costs
        // ~300 integer instructions per tile against the tile's 336 MFMAs per wave, and none of them hides behind the fp32 MFMA.
        const bool interior = d0 >= 1 && d0 + BTD < D && h0 >= 1 && h0 + BTH < H && w0 >= 1 && w0 + FTW < W && co_full;
        if (interior) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                xo[0][j] = xrel[0] + (unsigned)(j * x_ld);
                xo[1][j] = xrel[1] + (unsigned)(j * x_ld);
                yo[j] = yrel + (unsigned)(j * y_ld);
            }
        } else {
            bits = 0u;
#pragma unroll
            for (int u = 0; u < 2; ++u) {   // ---- X
                const int row = s_row + 16 * u;
                const int gd = d0 - 1 + row / BHH, gh = h0 - 1 + row % BHH;
                const bool rok = (unsigned)gd < (unsigned)D && (unsigned)gh < (unsigned)H;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bool ok = rok && w0 + 4 * s_wg + j < W;
                    xo[u][j] = ok ? xrel[u] + (unsigned)(j * x_ld) : xsafe;
                    bits |= ok ? 1u << (4 * u + j) : 0u;
                }
            }
            {   // ---- dY
                const int gd = d0 + s_row / BTH, gh = h0 + s_row % BTH;
                const bool rok = yrow_ok && gd < D && gh < H && cob * 16 + 4 * s_q < Co;       // host guarantees Co % 4 == 0
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bool ok = rok && w0 + 4 * s_wg + j < W;
                    yo[j] = ok ? yrel + (unsigned)(j * y_ld) : 0u;
                    bits |= ok ? 1u << (8 + j) : 0u;
                }
            }
            {   // ---- dY W-halo voxels w0 - 1 / w0 + 16
                const int gd = d0 + h_row / BTH, gh = h0 + h_row % BTH, gw = h_side ? w0 + FTW : w0 - 1;
                const bool ok = hrow_ok && gd < D && gh < H && (unsigned)gw < (unsigned)W && cob * 16 + 4 * h_q < Co;
                ho = ok ? hrel : (unsigned)y_ld;
                bits |= ok ? 1u << 12 : 0u;
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            vx[0][j] = ldg4(xb + xo[0][j]);
            vx[1][j] = ldg4(xb + xo[1][j]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) vy[j] = ldg4(yb + yo[j]);
        vh = ldg4(yb - y_ld + ho);
        okbits = bits;
        border = !interior;
    };
    auto store_tile_8 = [&]() {
        if (border) {   // wave-uniform; the loads landed long ago (a tile of MFMAs lies between load_tile and store_tile)
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (!((okbits >> (4 * u + j)) & 1u)) vx[u][j] = zero4;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (!((okbits >> (8 + j)) & 1u)) vy[j] = zero4;
            if (!((okbits >> 12) & 1u)) vh = zero4;
        }
        if (xon) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                char* dst = xs + ((s_row + 16 * u) * 16 + 4 * s_q) * DLS + 16 * rot_slot(s_wg, 4 * s_q);   // 4 channels share a rotation
                *reinterpret_cast<float4*>(dst) = make_float4(vx[u][0].x, vx[u][1].x, vx[u][2].x, vx[u][3].x);
                *reinterpret_cast<float4*>(dst + DLS) = make_float4(vx[u][0].y, vx[u][1].y, vx[u][2].y, vx[u][3].y);
                *reinterpret_cast<float4*>(dst + 2 * DLS) = make_float4(vx[u][0].z, vx[u][1].z, vx[u][2].z, vx[u][3].z);
                *reinterpret_cast<float4*>(dst + 3 * DLS) = make_float4(vx[u][0].w, vx[u][1].w, vx[u][2].w, vx[u][3].w);
            }
        }
        if (yon) {
            char* dst = ys + (s_row * 16 + 4 * s_q) * DLS + 16 * rot_slot(s_wg, 4 * s_q);
            *reinterpret_cast<float4*>(dst) = make_float4(vy[0].x, vy[1].x, vy[2].x, vy[3].x);
            *reinterpret_cast<float4*>(dst + DLS) = make_float4(vy[0].y, vy[1].y, vy[2].y, vy[3].y);
            *reinterpret_cast<float4*>(dst + 2 * DLS) = make_float4(vy[0].z, vy[1].z, vy[2].z, vy[3].z);
            *reinterpret_cast<float4*>(dst + 3 * DLS) = make_float4(vy[0].w, vy[1].w, vy[2].w, vy[3].w);
        }
        if (hon) {
            char* dst = yh + (h_row * 16 + 4 * h_q) * 8 + 4 * h_side;
            *reinterpret_cast<float*>(dst) = vh.x;
            *reinterpret_cast<float*>(dst + 8) = vh.y;
            *reinterpret_cast<float*>(dst + 16) = vh.z;
            *reinterpret_cast<float*>(dst + 24) = vh.w;
        }
    };

    auto load_tile = [&](int tile) {
        if constexpr (CI8 || CO8) load_tile_8(tile);
        else load_tile_base(tile);
    };
    auto store_tile = [&]() {
        if constexpr (CI8 || CO8) store_tile_8();
        else store_tile_base();
    };

    const TileWalk tw = tile_walk(ntiles);
    if (tw.count > 0) load_tile(tw.first);
    for (int k = 0; k < tw.count; ++k) {
        __syncthreads();   // the previous tile's MFMAs are done with the LDS tile
        store_tile();
        __syncthreads();
        load_tile(tw.first + (k + 1 < tw.count ? k + 1 : k) * tw.stride);   // (the last iteration re-reads its own tile: no branch)
        __builtin_amdgcn_sched_barrier(0);

        // ---- 3 output rows per wave x 9 (kd, kh) x 3 kw x 4 k-steps
        // the wave's three rows are consecutive in h (same d): row pointers advance by a constant, every tap is an
        // immediate offset — no per-row address arithmetic (each VALU instruction costs MFMA time)
        const char* xrow = xrow0;
        const char* yrow = yrow0;
        const char* ypl = ypl0;
        const char* ynr = ynr0;
#pragma unroll 1
        for (int r = 0; r < BYR / 4; ++r, xrow += 16 * DLS, yrow += 16 * DLS, ypl += pl_step, ynr += nr_step) {
            const float4 b = *reinterpret_cast<const float4*>(yrow);
            const float pl = *reinterpret_cast<const float*>(ypl);   // dY[first - 1]
            const float nr = *reinterpret_cast<const float*>(ynr);   // dY[last + 1]
            const float b0[4] = {b.y, b.z, b.w, nr}, b1[4] = {b.x, b.y, b.z, b.w}, b2[4] = {pl, b.x, b.y, b.z};   // dY[u+1], dY[u], dY[u-1]
            if (BIAS) {
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[TG] = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, b1[j], acc[TG], 0, 0, 0);
            }
            float bp[4];   // CO8: kw 0 (columns 0..7) | kw 1 (columns 8..15)
#pragma unroll
            for (int j = 0; j < 4; ++j) bp[j] = bhs ? b1[j] : b0[j];
#pragma unroll
            for (int pa = 0; pa < NA; ++pa) {
                // the line offset is a compile-time constant after unrolling (CI8: one per-lane register per pair)
                const float4 g = *reinterpret_cast<const float4*>(xrow + (CI8 ? aoff[pa] : ((pa / 3) * BHH + pa % 3) * 16 * DLS));
                const float a[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if constexpr (CO8) {
                        acc[pa * 2 + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], bp[j], acc[pa * 2 + 0], 0, 0, 0);
                        acc[pa * 2 + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b2[j], acc[pa * 2 + 1], 0, 0, 0);
                    } else {
                        acc[pa * 3 + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b0[j], acc[pa * 3 + 0], 0, 0, 0);
                        acc[pa * 3 + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b1[j], acc[pa * 3 + 1], 0, 0, 0);
                        acc[pa * 3 + 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b2[j], acc[pa * 3 + 2], 0, 0, 0);
                    }
                }
            }
        }
    }

    // combine the 4 waves in a fixed order through LDS, then one partial per workgroup
    __syncthreads();
    float* red = lds;  // [TGA][256]
    for (int w = 0; w < 4; ++w) {
        if (wv == w) {
#pragma unroll
            for (int t = 0; t < TGA; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int o = t * 256 + (4 * kq + r) * 16 + li;
                    red[o] = (w == 0) ? acc[t][r] : red[o] + acc[t][r];
                }
        }
        __syncthreads();
    }
    float* out = part + (((size_t)blockIdx.x * gridDim.y + cit) * gridDim.z + cob) * (TGA * 256);
    for (int i = tid; i < TGA * 256; i += 256) out[i] = red[i];
}

// dw[co][ci][tap] = sum_p part[p][cit][cob][tg][row][col]   (+ dbias[co] from the extra accumulator of cit == 0)
// Threads walk the partial layout itself (64 consecutive elements per wave => coalesced 256-byte reads of every
// partial), 4 partial-lanes per element combined through LDS in double; the (tiny) result is scattered into torch's
// (Co, Ci, 3,3,3) layout.
__global__ void __launch_bounds__(256)
wgrad_mfma_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, float* __restrict__ dbias, int P,
                         int CIT, int COB, int CK, int TG, int TGA, int Ci, int Co, int mode8) {
    __shared__ double red[256];
    const int el = threadIdx.x & 63, ql = threadIdx.x >> 6;
    const int nelem = CIT * COB * TGA * 256;
    const int e = blockIdx.x * 64 + el;
    const size_t pstride = (size_t)nelem;
    double s = 0.0;
    if (e < nelem) {
        double s1 = 0.0, s2 = 0.0, s3 = 0.0;   // four independent chains: more partial loads in flight
        int q = ql;
        for (; q + 12 < P; q += 16) {
            s += (double)part[(size_t)q * pstride + e];
            s1 += (double)part[(size_t)(q + 4) * pstride + e];
            s2 += (double)part[(size_t)(q + 8) * pstride + e];
            s3 += (double)part[(size_t)(q + 12) * pstride + e];
        }
        for (; q < P; q += 4) s += (double)part[(size_t)q * pstride + e];
        s = (s + s1) + (s2 + s3);
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if (ql != 0 || e >= nelem) return;
    s = red[el] + red[64 + el] + red[128 + el] + red[192 + el];
    const int col = e & 15, row = (e >> 4) & 15;
    int t = e >> 8;
    const int tg = t % TGA;
    t /= TGA;
    const int cob = t % COB, cit = t / COB;
    // mode8 (v6 with eight-channel operands; bit 0: Ci == 8, bit 1: Co == 8): rows / columns 8..15 repeat the channels for another tap
    const bool ci8 = mode8 & 1, co8 = mode8 & 2;
    const int co = co8 ? (col & 7) : cob * 16 + col;
    if (co >= Co) return;
    if (tg == TG) {  // bias accumulator (all rows equal): take row 0 of the first ci tile
        if (dbias != nullptr && cit == 0 && row == 0 && !(co8 && col >= 8)) dbias[co] = (float)s;
        return;
    }
    int ci, tap;
    if (mode8) {
        ci = ci8 ? (row & 7) : cit * 16 + row;
        tap = wg6_tap(ci8, co8, tg, row, col);
        if (tap < 0) return;
    } else if (CK == 16) { ci = cit * 16 + row; tap = tg; }
    else if (CK == 8) { ci = cit * 8 + (row & 7); tap = 2 * tg + (row >> 3); }
    else { ci = cit; tap = 16 * tg + row; }
    if (tap < 27 && ci < Ci) dw[((size_t)co * Ci + ci) * 27 + tap] = (float)s;
}

struct MfmaWgradPlan {
    int CK, CIT, COB, TG, P, tilesD, tilesH, tilesW, ntiles, v2, mode8, segl;
    size_t part_floats, smem;
};

static bool mfma_wgrad_plan(const Mri3dConvGeom& g, MfmaWgradPlan& p) {
    if (!(g.kd == 3 && g.kh == 3 && g.kw == 3 && g.sd == 1 && g.sh == 1 && g.sw == 1 && g.pd == 1 && g.ph == 1 &&
          g.pw == 1 && g.dd == 1 && g.dh == 1 && g.dw == 1))
        return false;
    // kernel: 0 = first-layer kernel (Cin = 1: 16 taps per M-tile), 1 = v3 (Cin % 8 == 0, register prefetch),
    // 2 = v4 (Cin % 16 == 0, small double-buffered tile: only for bf16 tensors the bf16 MFMA kernel cannot take),
    // 3 = bf16 MFMA (bf16 tensors, Cin % 8 == 0: a 16-channel ci-tile whose upper half may be empty),
    // 4 = v6 (fp32, Cin % 16 == 0, transposed tile: 110 / 104 / 102 TFLOP/s on 48->16 / 96->32 / 16->16 against v4's 103 / 90 / 99)
    p.v2 = (g.ci % 8 == 0 && g.co % 4 == 0 && g.y_ld % 4 == 0) ? (g.ci % 16 == 0 ? 2 : 1) : 0;
    if (g.dtype == MRI3D_BF16 && g.ci % 8 == 0 && g.co % 8 == 0 && g.x_ld % 8 == 0 && g.y_ld % 8 == 0) p.v2 = 3;
    if (p.v2 == 2 && g.dtype == MRI3D_F32) p.v2 = 4;
    // v6 with eight-channel operands: 16k -> 8 (dY columns paired over kw) and 8 -> 8 (X rows paired over (kd, kh) as well)
    p.mode8 = 0;
    if (g.dtype == MRI3D_F32 && g.co == 8 && g.y_ld % 4 == 0 && g.x_ld % 4 == 0 && (g.ci % 16 == 0 || g.ci == 8)) {
        p.v2 = 4;
        p.mode8 = 2 | (g.ci == 8 ? 1 : 0);
    }
#ifndef MRI3D_BF16_WGRAD_MARCH_MIN_D
#define MRI3D_BF16_WGRAD_MARCH_MIN_D 8   // (tuning builds: a huge value keeps every bf16 layer on the tile kernel)
#endif
    // bf16, marching along d (conv_mfma_wgrad_bf16t_kernel): when the columns x segments give every workgroup at least three
    // tasks — segments of 40 planes, or 20 for smaller volumes (each task pays 4 staging-only fill steps; with 10-plane segments
    // the 32 -> 32 layer at 80x96x80 ran 0.132 ms against the tile kernel's 0.122)
    p.segl = 0;
    if (p.v2 == 3 && g.di >= MRI3D_BF16_WGRAD_MARCH_MIN_D) {
        const int pairs5 = cdiv(g.ci, 16) * cdiv(g.co, 16);
        const int P5 = std::max(1, 512 / std::max(1, pairs5));
        const int64_t cols = (int64_t)g.n * cdiv(g.hi, MTH) * cdiv(g.wi, BTW);
        for (int sl = kMarchSeg; sl >= 20 && p.segl == 0; sl /= 2)
            if (cols * cdiv(g.di, sl) >= (int64_t)3 * P5) p.segl = sl;
        if (p.segl) p.v2 = 5;
    }
    // (The same structure for fp32 — tools/experiments/wgrad6m_kernel.hip, parity-green — measured no gain: 16 -> 16 1.20 -> 1.25 ms,
    // 48 -> 16 3.35 -> 3.40 ms, 96 -> 32 1.83 -> 1.81 ms, only 32^3 x 512 patches 2.21 -> 2.03 ms.  The fp32 kernel is MFMA-bound and
    // at its register limit; fewer staged bytes buy it nothing.)
    if (p.v2 == 3 || p.v2 == 5) p.CK = 16;
    else if (p.mode8) p.CK = 16;
    else if (p.v2 != 0 && g.ci % 16 == 0) p.CK = 16;
    else if (p.v2 != 0) p.CK = 8;
    else if (g.ci == 1) {
        // Conv3d(1, 8, 3) and the 1 -> 1 stencil have direct kernels in conv_generic.hip (conv_cin1_wgrad_kernel: 0.25 vs
        // 0.47 ms on 2 x 160x192x160; conv_c1c1_wgrad_kernel); Co = 16 stays here (0.22 vs 0.24 ms on 16 x 64^3)
        if ((g.co == 8 && g.y_ld % 4 == 0) || g.co == 1) return false;
        p.CK = 1;
    } else return false;
    if (p.CK >= 4 && g.x_ld % 4 != 0) return false;
    p.CIT = cdiv(g.ci, p.CK);
    p.COB = cdiv(g.co, 16);
    p.TG = p.mode8 ? wg6_groups(p.mode8 & 1, p.mode8 & 2) : wg_tap_groups(p.CK);
    p.tilesD = cdiv(g.di, p.v2 >= 3 ? BTD : WTD);
    p.tilesH = cdiv(g.hi, p.v2 >= 3 ? BTH : (p.v2 == 2 ? V4TH : WTH));
    p.tilesW = cdiv(g.wi, p.v2 == 3 ? BTW : (p.v2 == 4 ? FTW : WTW));
    if (p.v2 == 5) {   // tasks = (sample, segment of d, column): tilesD holds the segments
        p.tilesD = cdiv(g.di, p.segl);
        p.tilesH = cdiv(g.hi, MTH);
        p.tilesW = cdiv(g.wi, BTW);
    }
    int64_t nt = (int64_t)g.n * p.tilesD * p.tilesH * p.tilesW;
    if (nt > 0x7fffffff) return false;
    p.ntiles = (int)nt;
    int pairs = p.CIT * p.COB;
    if (pairs > 65535) return false;
    int P = 512 / pairs;  // ~2 resident workgroups per CU in total
    if (P < 1) P = 1;
    if (P > p.ntiles) P = p.ntiles;
    p.P = P;
    p.part_floats = (size_t)P * p.CIT * p.COB * (p.TG + 1) * 256;
    size_t xs = ((size_t)WHVOX * p.CK + 3) & ~(size_t)3;
    size_t red = (size_t)(p.TG + 1) * 256;
    size_t tile_floats = xs + (size_t)WVOX * 16;
    p.smem = (tile_floats > red ? tile_floats : red) * sizeof(float);
    if (p.v2 == 1) {
        size_t xbuf = (size_t)((WHVOX * (p.CK / 4) + 255) / 256) * 256 * 4;
        p.smem = std::max<size_t>(xbuf + (size_t)WVOX * 16, red) * sizeof(float);
    } else if (p.v2 == 2) {
        p.smem = std::max<size_t>((size_t)2 * V4XBUF + 2 * V4YBUF, red) * sizeof(float);
    } else if (p.v2 == 5) {
        p.smem = std::max<size_t>((size_t)TLDS, red * sizeof(float));
    } else if (p.v2 >= 3) {
        p.smem = std::max<size_t>((size_t)BXS + BYS + BYH, red * sizeof(float));
    }
    return true;
}

bool conv_mfma_cat_supported(const Mri3dConvGeom& g, int split, int second_ld, int pass) {
    const bool bf = g.dtype == MRI3D_BF16;
    if (split <= 0 || split >= g.ci || split % 16 != 0 || second_ld % (bf ? 8 : 4) != 0 || second_ld < g.ci - split) return false;
    if (direct_only(g)) return false;
    MfmaFwdPlan p;
    MfmaWgradPlan q;
    if (pass == MRI3D_PASS_FWD) return mfma_fwd_plan(g, false, p) && p.small != 1 && (g.ci - split) % 8 == 0;
    if (pass == MRI3D_PASS_DGRAD) return mfma_fwd_plan(g, true, p) && p.small != 1 && (g.ci - split) % 4 == 0;   // N side: 16-channel tiles
    if (pass == MRI3D_PASS_WGRAD) return mfma_wgrad_plan(g, q) && (q.v2 == 4 ? (g.ci - split) % 16 == 0 : (q.v2 == 3 || q.v2 == 5));
    return false;
}

bool conv_mfma_supported(const Mri3dConvGeom& g, int pass) {
    MfmaFwdPlan p;
    MfmaWgradPlan q;
    DirectPlan dp;
    if (pass != MRI3D_PASS_WGRAD && direct_only(g)) return direct_plan(g, pass == MRI3D_PASS_DGRAD, dp);
    if (pass == MRI3D_PASS_FWD) return mfma_fwd_plan(g, false, p);
    if (pass == MRI3D_PASS_DGRAD) return mfma_fwd_plan(g, true, p);
    if (pass == MRI3D_PASS_WGRAD) return mfma_wgrad_plan(g, q);
    return false;
}

size_t conv_mfma_workspace_bytes(const Mri3dConvGeom& g, int pass) {
    MfmaFwdPlan p;
    MfmaWgradPlan q;
    DirectPlan dp;
    if (pass != MRI3D_PASS_WGRAD && direct_only(g)) return direct_plan(g, pass == MRI3D_PASS_DGRAD, dp) ? dp.wp_floats * sizeof(float) : 0;
    if (pass == MRI3D_PASS_FWD && mfma_fwd_plan(g, false, p))
        return std::max(std::max(p.wp_floats, p.s_wp_floats) * sizeof(float), conv_march_workspace_bytes(g, false));
    if (pass == MRI3D_PASS_DGRAD && mfma_fwd_plan(g, true, p))
        return std::max(std::max(p.wp_floats, p.s_wp_floats) * sizeof(float), conv_march_workspace_bytes(g, true));
    if (pass == MRI3D_PASS_WGRAD && mfma_wgrad_plan(g, q)) return q.part_floats * sizeof(float);
    return 0;
}

// set a kernel's dynamic-LDS limit once (not per launch)
#define MRI3D_SET_SMEM_ONCE(kern, bytes)                                                                              \
    do {                                                                                                              \
        static const hipError_t attr_ = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                      \
                                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)); \
        (void)attr_;                                                                                                  \
    } while (0)

template <typename T>
static void launch_mfma_wgrad_cin1(const MfmaWgradPlan& p, const Mri3dConvGeom& g, const T* x, const T* dy,
                                   float* part, bool bias, hipStream_t s) {
    dim3 grid(p.P, p.CIT, p.COB);
    if (bias) {
        hipLaunchKernelGGL((conv_mfma_wgrad_kernel<T, 1, true>), grid, dim3(256), p.smem, s, x, dy, part, g.n, g.di, g.hi, g.wi,
                           g.ci, g.x_ld, g.co, g.y_ld, p.tilesD, p.tilesH, p.tilesW, p.ntiles);
    } else {
        hipLaunchKernelGGL((conv_mfma_wgrad_kernel<T, 1, false>), grid, dim3(256), p.smem, s, x, dy, part, g.n, g.di, g.hi, g.wi,
                           g.ci, g.x_ld, g.co, g.y_ld, p.tilesD, p.tilesH, p.tilesW, p.ntiles);
    }
}

template <typename T>
static void run_mfma_wgrad(const MfmaWgradPlan& p, const Mri3dConvGeom& g, const T* x, const T* dy, float* part, bool bias,
                           hipStream_t s, ConvSplit sp = ConvSplit{nullptr, 0, 0}) {
    // the kernel template reserves the bias accumulator slot in the partial layout (TGA = TG + 1) only when BIAS
    if (p.v2 == 5) {
        if constexpr (sizeof(T) == 2) {
            dim3 grid(p.P, p.CIT, p.COB);
#define MRI3D_WGM(Bv)                                                                                                 \
    {                                                                                                                 \
        auto kern = conv_mfma_wgrad_bf16t_kernel<Bv>;                                                                 \
        MRI3D_SET_SMEM_ONCE(kern, p.smem);                                                                            \
        hipLaunchKernelGGL(kern, grid, dim3(256), p.smem, s, x, dy, part, g.n, g.di, g.hi, g.wi, g.ci, g.x_ld, g.co,  \
                           g.y_ld, p.tilesD, p.tilesH, p.tilesW, p.ntiles, (const bf16_t*)sp.second, sp.second_ld,    \
                           sp.split, p.segl);                                                                         \
    }
            if (bias) MRI3D_WGM(true) else MRI3D_WGM(false)
#undef MRI3D_WGM
        }
    } else if (p.v2 == 3) {
        if constexpr (sizeof(T) == 2) {
            dim3 grid(p.P, p.CIT, p.COB);
#define MRI3D_WGB(Bv)                                                                                                 \
    {                                                                                                                 \
        auto kern = conv_mfma_wgrad_bf16_kernel<Bv>;                                                                  \
        MRI3D_SET_SMEM_ONCE(kern, p.smem);                                                                            \
        hipLaunchKernelGGL(kern, grid, dim3(256), p.smem, s, x, dy, part, g.n, g.di, g.hi, g.wi, g.ci, g.x_ld, g.co,  \
                           g.y_ld, p.tilesD, p.tilesH, p.tilesW, p.ntiles, (const bf16_t*)sp.second, sp.second_ld,    \
                           sp.split);                                                                                 \
    }
            if (bias) MRI3D_WGB(true) else MRI3D_WGB(false)
#undef MRI3D_WGB
        }
    } else if (p.v2 == 4) {
        if constexpr (sizeof(T) == 4) {
            dim3 grid(p.P, p.CIT, p.COB);
#define MRI3D_WG6(Bv)                                                                                                 \
    if (p.mode8 == 3) MRI3D_WG6M(Bv, true, true) else if (p.mode8 == 2) MRI3D_WG6M(Bv, false, true) else if (p.mode8 == 1) MRI3D_WG6M(Bv, true, false) else MRI3D_WG6M(Bv, false, false)
#define MRI3D_WG6M(Bv, I8, O8)                                                                                        \
    {                                                                                                                 \
        auto kern = conv_mfma_wgrad6_kernel<Bv, I8, O8>;                                                              \
        MRI3D_SET_SMEM_ONCE(kern, p.smem);                                                                            \
        hipLaunchKernelGGL(kern, grid, dim3(256), p.smem, s, x, dy, part, g.n, g.di, g.hi, g.wi, g.ci, g.x_ld, g.co,  \
                           g.y_ld, p.tilesD, p.tilesH, p.tilesW, p.ntiles, (const float*)sp.second, sp.second_ld,     \
                           sp.split);                                                                                 \
    }
            if (bias) { MRI3D_WG6(true) } else { MRI3D_WG6(false) }
#undef MRI3D_WG6
#undef MRI3D_WG6M
        }
    } else if (p.v2 == 2) {
      if constexpr (sizeof(T) == 2) {   // fp32 tensors with Cin % 16 == 0 always take v6
        dim3 grid(p.P, p.CIT, p.COB);
#define MRI3D_WG4(Bv)                                                                                                 \
    {                                                                                                                 \
        auto kern = conv_mfma_wgrad4_kernel<T, Bv>;                                                                   \
        MRI3D_SET_SMEM_ONCE(kern, p.smem);                                                                            \
        hipLaunchKernelGGL(kern, grid, dim3(256), p.smem, s, x, dy, part, g.n, g.di, g.hi, g.wi, g.ci, g.x_ld, g.co,  \
                           g.y_ld, p.tilesD, p.tilesH, p.tilesW, p.ntiles);                                           \
    }
        if (bias) MRI3D_WG4(true) else MRI3D_WG4(false)
#undef MRI3D_WG4
      }
    } else if (p.v2 == 1) {
        dim3 grid(p.P, p.CIT, p.COB);
#define MRI3D_WG3(CKv, Bv)                                                                                            \
    {                                                                                                                 \
        auto kern = conv_mfma_wgrad3_kernel<T, CKv, Bv>;                                                              \
        MRI3D_SET_SMEM_ONCE(kern, p.smem);                                                                            \
        hipLaunchKernelGGL(kern, grid, dim3(256), p.smem, s, x, dy, part, g.n, g.di, g.hi, g.wi, g.ci, g.x_ld, g.co,  \
                           g.y_ld, p.tilesD, p.tilesH, p.tilesW, p.ntiles);                                           \
    }
        if (bias) MRI3D_WG3(8, true)
        else MRI3D_WG3(8, false)
#undef MRI3D_WG3
    } else {
        launch_mfma_wgrad_cin1<T>(p, g, x, dy, part, bias, s);
    }
}

static int run_wgrad(const Mri3dConvGeom& g, const void* x, const void* dy, float* dw, float* dbias, void* ws, size_t ws_bytes,
                     hipStream_t s, ConvSplit sp);

int conv_mfma_wgrad(const Mri3dConvGeom& g, const void* x, const void* dy, float* dw, float* dbias, void* ws,
                    size_t ws_bytes, hipStream_t s) {
    return run_wgrad(g, x, dy, dw, dbias, ws, ws_bytes, s, ConvSplit{nullptr, 0, 0});
}

int conv_mfma_wgrad_cat(const Mri3dConvGeom& g, const void* x, const void* x2, int split, int x2_ld, const void* dy, float* dw,
                        float* dbias, void* ws, size_t ws_bytes, hipStream_t s) {
    return run_wgrad(g, x, dy, dw, dbias, ws, ws_bytes, s, ConvSplit{x2, split, x2_ld});
}

static int run_wgrad(const Mri3dConvGeom& g, const void* x, const void* dy, float* dw, float* dbias, void* ws, size_t ws_bytes,
                     hipStream_t s, ConvSplit sp) {
    MfmaWgradPlan p;
    MRI3D_REQUIRE(mfma_wgrad_plan(g, p), MRI3D_ENOTSUP, "conv3d_wgrad(mfma): unsupported geometry");
    MRI3D_REQUIRE(ws && ws_bytes >= p.part_floats * sizeof(float), MRI3D_EWORKSPACE,
                  "conv3d_wgrad(mfma): workspace %zu < %zu", ws_bytes, p.part_floats * sizeof(float));
    MRI3D_REQUIRE(aligned_vec4(g.dtype, x, dy), MRI3D_EINVAL, "conv3d_wgrad(mfma): x/dy must be aligned to 4 elements");
    MRI3D_REQUIRE((p.v2 != 3 && p.v2 != 5) || ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy)) & 15) == 0, MRI3D_EINVAL,
                  "conv3d_wgrad(bf16 mfma): x/dy must be 16-byte aligned");
    float* part = static_cast<float*>(ws);
    const bool bias = dbias != nullptr;
    MRI3D_REQUIRE(sp.second == nullptr || p.v2 >= 3, MRI3D_ENOTSUP, "conv3d_wgrad(mfma): split operands need the transposed-tile kernels");
    MRI3D_DISPATCH_DTYPE(g.dtype, T, { run_mfma_wgrad<T>(p, g, static_cast<const T*>(x), static_cast<const T*>(dy), part, bias, s, sp); });
    const int TGA = p.TG + (bias ? 1 : 0);
    const int nelem = p.CIT * p.COB * TGA * 256;
    hipLaunchKernelGGL(wgrad_mfma_reduce_kernel, dim3(cdiv(nelem, 64)), dim3(256), 0, s, part, dw, dbias, p.P, p.CIT,
                       p.COB, p.CK, p.TG, TGA, g.ci, g.co, p.mode8);
    return check_launch("conv3d_wgrad(mfma)");
}

}  // namespace mri3d
