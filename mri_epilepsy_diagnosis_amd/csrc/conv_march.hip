// conv_march.hip — Conv3d 3x3x3 / stride 1 / pad 1 forward and data gradient as a MARCH ALONG d (round 3).
//
// Replaces, for the layers its plan accepts, the tiled kernel of conv_mfma.hip (nn.Conv3d forward / dgrad of unet.UNet's
// full-resolution layers, /root/reference segmentation/routine.py:346-356; Modified3DUNet, modified_3dunet.py:97-189).
//
// Why: the tiled kernel stages the 6 x 10 x 18 halo of a 4 x 8 x 16 output tile per 32-byte channel chunk — every input byte
// passes the CU's memory pipeline 2.1 times and the staging (not the MFMAs) bounds the bf16 layers (MFMA busy
// 36-45 %).  Here a WAVE owns a column of 8 rows x 16 voxels and marches through a segment of d planes:
//
//   * scatter form: input plane p contributes to the three output planes p+1, p, p-1 (kd = 0, 1, 2), whose accumulators
//     (3 x 8 row tiles x 4 VGPRs) stay in registers; an output plane is stored when input plane p+1 has been consumed.  Each
//     input plane (10 x 18 halo voxels x one 32-byte channel chunk = 5.6 KB) is staged ONCE per column: 1.41x instead of 2.11x.
//   * LDS holds only the plane chunk being multiplied and the one(s) in flight: 2 (or 3) x 6 KB per wave, filled by LDS-DMA
//     (buffer_load ... lds; zero fill outside the volume by range check).  The buffers are PRIVATE to the wave: no workgroup
//     barrier anywhere in the march — a wave waits only for its own DMA pieces (counted vmcnt), and the eight waves of a CU drift
//     apart so that one wave's wait sits under another's MFMAs.
//   * the voxel fragments of an input plane do not depend on kd: the 10 row fragments of the (kw = 0 | kw = 1) pairing serve
//     the nine (kd, kh) groups, 8 + 8 more serve the kw = 2 taps: 26 ds_read per 120 MFMAs (tiled kernel: 56 per 112).
//   * the 27 taps are packed without a zero slot: per kd four PAIRED groups (two taps x 16 bf16 channels = one K = 32 MFMA)
//     and one SINGLE tap on the K = 16 MFMA (fp32: two instead of four K = 4 steps).
//   * weights (<= 55 KB per 16-channel output block) are copied to LDS once per workgroup and read from there, so the vector
//     memory queue carries nothing but the plane pieces and the output stores.
//
// Roofline: bf16 layers are bound by HBM bytes; algorithmic bytes per launch = input + output tensors once.
#include "common.h"
#include "mfma_util.h"
#include <algorithm>
#include <type_traits>
#include <utility>

namespace mri3d {

constexpr int MH = 8, MW = 16;              // a wave's output column: rows x voxels
constexpr int MHH = MH + 2, MHW = MW + 2;   // halo plane 10 x 18
constexpr int MVOX = MHH * MHW;             // 180 voxels
constexpr int MPIECES = 6;                  // 1-KiB DMA wave-instructions per plane chunk (360 of 384 16-byte pieces used)
constexpr int MBUF = MVOX * 32;             // bytes per LDS plane buffer (5760): the sixth piece is issued for lanes 0..39 only
constexpr int MWKD = 4 * 1024 + 512;        // packed weight bytes per (chunk, kd): four paired groups + the single tap
constexpr int MWCHUNK = 3 * MWKD;           // 13,824
constexpr int MROW = MHW * 32;              // LDS bytes per halo row (576)

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));

// Packed weights of the marching kernel: image [N-block nb][chunk][kd][group g][lane][elements]
//   g = 0..2 (paired, kh = g):  k-slot ks = lane >> 5 carries tap (kd, kh, kw = ks);  g = 3 (paired): tap (kd, kh = ks, kw = 2)
//       element s (PE per lane: 8 bf16 / 4 fp32) = channel chunk*CK + PE*((lane >> 4) & 1) + s
//   g = 4 (single tap (kd, 2, 2)): element s (PE/2 per lane) = channel chunk*CK + (PE/2)*(lane >> 4) + s
//   weight row rho = lane & 15 carries output channel nc = nb*16 + 4*sigma(rho >> 2) + (rho & 3), sigma = (0, 2, 1, 3): the MFMA
//   leaves row rho in lane group kq = rho >> 2, so lane groups kq and kq + 2 (lanes 32 apart) hold EIGHT consecutive channels of
//   a voxel — one v_permlane32_swap per dword then gives every lane 16 contiguous bytes of a bf16 row (cdna_hip_programming.md T21)
//   forward: W[nc][kc][tap];  data gradient: W[kc][nc][26 - tap]
template <typename WT>
__global__ void pack_w_march_kernel(const float* __restrict__ w, WT* __restrict__ wp, int Co, int Ci, int dgrad, int nchunks,
                                    int NTT) {
    constexpr int PE = 16 / sizeof(WT), CK = 2 * PE;
    constexpr int PER_KD = 288 * PE;   // elements per (chunk, kd): 4 x 64 x PE + 64 x PE/2
    const int total = NTT * nchunks * 3 * PER_KD;
    const int Kc = dgrad ? Co : Ci, Nc = dgrad ? Ci : Co;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        int e = i % PER_KD, t = i / PER_KD;
        const int kd = t % 3;
        t /= 3;
        const int chunk = t % nchunks, nb = t / nchunks;
        int kh, kw, kc, lane;
        if (e < 256 * PE) {
            const int g = e / (64 * PE), s = e % PE;
            lane = (e / PE) & 63;
            const int ks = lane >> 5;
            kh = g < 3 ? g : ks;
            kw = g < 3 ? ks : 2;
            kc = chunk * CK + PE * ((lane >> 4) & 1) + s;
        } else {
            e -= 256 * PE;
            lane = e / (PE / 2);
            kh = 2, kw = 2;
            kc = chunk * CK + (PE / 2) * (lane >> 4) + e % (PE / 2);
        }
        const int rho = lane & 15, sig = ((rho >> 2) & 1) * 2 + (rho >> 3);
        const int nc = nb * 16 + 4 * sig + (rho & 3), tap = (kd * 3 + kh) * 3 + kw;
        float v = 0.f;
        if (nc < Nc && kc < Kc) v = dgrad ? w[((size_t)kc * Ci + nc) * 27 + (26 - tap)] : w[((size_t)nc * Ci + kc) * 27 + tap];
        wp[i] = (WT)v;
    }
}

#if defined(MRI3D_EXPERIMENT_STAMPS)   // tuning builds: s_memtime phase sums of wave 0 of workgroup 0 (tools/march_bench.py --stamps)
__device__ unsigned long long g_march_stamps[8];
__device__ unsigned long long g_march_span[4 * 1024];   // per workgroup of the LAST launch: 100 MHz ticks at kernel entry, march start (wave 0), march end (wave 0), kernel end
extern "C" void mri3d_debug_march_spans(unsigned long long* out) {
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_march_span), sizeof(unsigned long long) * 4 * 1024);
}
extern "C" void mri3d_debug_march_stamps(unsigned long long* out, int reset) {
    if (reset) {
        unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_march_stamps), z, sizeof(z));
    } else {
        (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_march_stamps), sizeof(unsigned long long) * 8);
    }
}
#define MARCH_STAMP(var)                                   \
    __builtin_amdgcn_sched_barrier(0);                     \
    const unsigned long long var = __builtin_readcyclecounter(); \
    __builtin_amdgcn_sched_barrier(0)
#define MARCH_STAMP_ADD(slot, a, b) (stamp_acc[slot] += (b) - (a))
#else
#define MARCH_STAMP(var)
#define MARCH_STAMP_ADD(slot, a, b)
#endif

// The 24 accumulators of a wave (3 output planes x 8 rows x one 16x16 tile) live in the registers v[160:255] BY NAME:
// accumulator (slot s, row m) = v[160 + 4*(8*s + m) .. + 3].  The MFMAs, the conversions / sums of the store path and the initial
// clear are inline asm on those physical registers; hipcc's allocator never sees them and is capped at v0..v159 for its own values
// (amdgpu_num_vgpr on the kernel).  Left to the allocator — builtins, or asm with tied "+v" operands — it kept splitting the 96
// loop-carried registers: a copy of every accumulator per item, or spills.  Kept in the accumulation file (a[0:95]) they are as
// stable, but every value stored costs a v_accvgpr_read first, and vector instructions are what the march has least of: beside two
// waves of back-to-back MFMAs a plain VALU instruction issues every 14-28 cycles (measured: 75 of them per plane took 1 100-2 100).
// hipcc pads nothing around inline asm (cdna_hip_programming.md §5.7): two MFMAs on one accumulator are at least 8 MFMAs apart in
// every phase; VALU reads of accumulators next to MFMAs are fenced by acc_fence().
template <int... I, typename F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(std::make_integer_sequence<int, N>{}, f); }

constexpr int ACC0 = 160;    // first accumulator register
#define MRI3D_ACC_CLOBBERS                                                                                                       \
    "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167", "v168", "v169", "v170", "v171", "v172", "v173", "v174", "v175",  \
        "v176", "v177", "v178", "v179", "v180", "v181", "v182", "v183", "v184", "v185", "v186", "v187", "v188", "v189", "v190",   \
        "v191", "v192", "v193", "v194", "v195", "v196", "v197", "v198", "v199", "v200", "v201", "v202", "v203", "v204", "v205",   \
        "v206", "v207", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219", "v220",   \
        "v221", "v222", "v223", "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235",   \
        "v236", "v237", "v238", "v239", "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247", "v248", "v249", "v250",   \
        "v251", "v252", "v253", "v254", "v255"

__device__ __forceinline__ void acc_fence() { asm volatile("s_nop 15\n\ts_nop 7"); }
// A = register offset of the accumulator inside the block: 4 * (8 * slot + row)
template <int A> __device__ __forceinline__ void acc_bf16_k32(const f32x4& w, const f32x4& f) {
    asm volatile("v_mfma_f32_16x16x32_bf16 v[%c2:%c3], %0, %1, v[%c2:%c3]" ::"v"(w), "v"(f), "n"(ACC0 + A), "n"(ACC0 + A + 3));
}
template <int A> __device__ __forceinline__ void acc_bf16_k32_first(const f32x4& w, const f32x4& f) {   // C = 0: the accumulator is re-used
    asm volatile("v_mfma_f32_16x16x32_bf16 v[%c2:%c3], %0, %1, 0" ::"v"(w), "v"(f), "n"(ACC0 + A), "n"(ACC0 + A + 3));
}
template <int A> __device__ __forceinline__ void acc_bf16_k16(const f32x2& w, const f32x2& f) {
    asm volatile("v_mfma_f32_16x16x16_bf16 v[%c2:%c3], %0, %1, v[%c2:%c3]" ::"v"(w), "v"(f), "n"(ACC0 + A), "n"(ACC0 + A + 3));
}
template <int A> __device__ __forceinline__ void acc_f32_k4(float w, float f) {
    asm volatile("v_mfma_f32_16x16x4_f32 v[%c2:%c3], %0, %1, v[%c2:%c3]" ::"v"(w), "v"(f), "n"(ACC0 + A), "n"(ACC0 + A + 3));
}
template <int A> __device__ __forceinline__ void acc_f32_k4_first(float w, float f) {
    asm volatile("v_mfma_f32_16x16x4_f32 v[%c2:%c3], %0, %1, 0" ::"v"(w), "v"(f), "n"(ACC0 + A), "n"(ACC0 + A + 3));
}
// store path, straight from the named registers.  R = register offset of an element PAIR (A, A + 2)
template <int R> __device__ __forceinline__ unsigned acc_cvt_pk_bf16() {   // (bf16(v[R]), bf16(v[R+1])), round to nearest even
    unsigned r;
    asm volatile("v_cvt_pk_bf16_f32 %0, v%c1, v%c2" : "=v"(r) : "n"(ACC0 + R), "n"(ACC0 + R + 1));
    return r;
}
template <int R> __device__ __forceinline__ f32x2 acc_pk_add(const f32x2& b) {   // v[R:R+1] + b
    f32x2 r;
    asm volatile("v_pk_add_f32 %0, v[%c1:%c2], %3" : "=v"(r) : "n"(ACC0 + R), "n"(ACC0 + R + 1), "v"(b));
    return r;
}
template <int R> __device__ __forceinline__ void acc_stat(f32x2& s1, f32x2& s2) {   // s1 += v[R:R+1], s2 += v[R:R+1]^2
    asm volatile("v_pk_add_f32 %0, %0, v[%c2:%c3]\n\tv_pk_fma_f32 %1, v[%c2:%c3], v[%c2:%c3], %1" : "+v"(s1), "+v"(s2) : "n"(ACC0 + R), "n"(ACC0 + R + 1));
}
template <int A> __device__ __forceinline__ void acc_store4(float* p) {   // the four registers of an accumulator as one 16-byte store
    asm volatile("global_store_dwordx4 %0, v[%c1:%c2], off" ::"v"(p), "n"(ACC0 + A), "n"(ACC0 + A + 3) : "memory");
}

struct MarchGeom {
    int N, D, H, W, Kc, in_ld, Nc, out_ld;
    int nchunks, NTT;      // 32-byte input chunks; 16-channel output blocks (one per workgroup)
    int wgh, wgw;          // the 8 wave-columns of a workgroup form wgh x wgw columns (h x w)
    int gch, gcw;          // workgroup-columns along h and w
    int nseg, seglen;      // d segments of seglen planes
    int nbuf;              // LDS plane buffers per wave (2 or 3)
    int waves;             // waves (columns) per workgroup: 8 or 12
    int stagger;           // waves 4..7 start this many 64-cycle sleeps late (0: none)
    int ksplit, in2_ld;    // forward over cat((x, x2)): input channels >= ksplit live in x2 (pitch in2_ld); 0: one tensor
    int nsplit, out2_ld;   // its data gradient: output channels >= nsplit go to y2; 0: one tensor
};

// WV = waves (columns) per workgroup
template <typename T, bool STATS, bool BIAS, int WV>
__device__ __forceinline__ void conv_march_body(const T* __restrict__ x, const T* __restrict__ x2, const unsigned char* __restrict__ wp,
                                                const float* __restrict__ bias, T* __restrict__ y, T* __restrict__ y2,
                                                double* __restrict__ stat_part, const MarchGeom& q) {
    constexpr int MWAVES = WV;
#if defined(MRI3D_EXPERIMENT_STAMPS)
    const unsigned long long span_t0 = __builtin_amdgcn_s_memrealtime();
#endif
    constexpr bool kBf16 = sizeof(T) == 2;
    constexpr int CK = 32 / sizeof(T), PE = 16 / sizeof(T);
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kq = lane >> 4;

    // workgroup -> (N-block, workgroup-column, segment, sample): consecutive ids live on one XCD (xcd_remap), and there the
    // N-blocks of a column (which stage the same input) and its w / h neighbours (which share halo voxels) run side by side
    int L = xcd_remap(blockIdx.x, gridDim.x);
    const int nb = L % q.NTT;
    L /= q.NTT;
    const int cw = L % q.gcw;
    L /= q.gcw;
    const int ch = L % q.gch;
    L /= q.gch;
    const int seg = L % q.nseg, n = L / q.nseg;

    // weights of this N-block -> LDS (all eight waves), statistics slots zeroed
    const int wbytes = q.nchunks * MWCHUNK;
    {
        const unsigned char* wsrc = wp + (size_t)nb * wbytes;
        for (int i = tid * 16; i < wbytes; i += WV * 64 * 16) *reinterpret_cast<uint4*>(smem + i) = *reinterpret_cast<const uint4*>(wsrc + i);
    }
    const int bufbase = (wbytes + 1023) & ~1023;
    unsigned char* const mybuf = smem + bufbase + wv * q.nbuf * MBUF;
    double* const stat_lds = reinterpret_cast<double*>(smem + bufbase + MWAVES * q.nbuf * MBUF);   // [8 waves][16 channels][2]
    float* const bias_lds = reinterpret_cast<float*>(stat_lds + MWAVES * 32);   // the N-block's 16 bias values (zeros beyond Nc)
    if constexpr (STATS) {
        if (tid < MWAVES * 32) stat_lds[tid] = 0.0;
    }
    if constexpr (BIAS) {
        if (tid < 16) bias_lds[tid] = nb * 16 + tid < q.Nc ? bias[nb * 16 + tid] : 0.f;
    }
    __syncthreads();

    const int wy = wv / q.wgw, wx = wv - wy * q.wgw;
    const int h0 = (ch * q.wgh + wy) * MH, w0 = (cw * q.wgw + wx) * MW;
    const int dlo = seg * q.seglen, dhi = min(dlo + q.seglen, q.D);
    const bool active = h0 < q.H && w0 < q.W && dlo < q.D;

    if (active) {
        // The two waves of a SIMD (w and w + 4) run the same program on equal columns and would stay in lock step: both in their
        // MFMA phase (sharing the pipe), then both in their store / wait phase (pipe idle).  Starting waves 4..7 half an item late
        // puts one wave's stores under the other's MFMAs (MI355X_MICROARCH.md, "Two waves per SIMD", item 9).
        if (wv >= 4)
            for (int i = 0; i < q.stagger * (wv >> 2); ++i) __builtin_amdgcn_s_sleep(1);
        const int D = q.D, H = q.H, W = q.W;
        const int plo = max(dlo - 1, 0), phi = min(dhi, D - 1);   // input planes this column consumes
        const int nitems = (phi - plo + 1) * q.nchunks;
        const int PD = q.nbuf - 1;                                // items of DMA lead

        // per-lane DMA geometry, fixed for the column: piece j of the lane = 16-byte piece j*64 + lane of the [voxel][32 B] image.
        // One input tensor of whole chunks (every layer but the decoder's cat((skip, upsampled)) convolution and the 8-channel
        // first bf16 chunk): the byte offset of every piece, the out-of-volume value folded in, is a per-lane CONSTANT of the
        // march — no vector instruction per piece (beside back-to-back MFMAs each one costs 14-28 cycles).  Otherwise vof holds the
        // voxel index and the offset is formed per piece from the chunk's tensor pitch.
        const bool fastdma = x2 == nullptr && q.Kc % CK == 0;   // wave-uniform
        unsigned vof[MPIECES], okmask = 0;
        const unsigned pieceb = (unsigned)(lane & 1) * 16u;
#pragma unroll
        for (int j = 0; j < MPIECES; ++j) {
            const int v = (j * 64 + lane) >> 1;
            const int row = v / MHW, col = v - row * MHW;
            const int gh = h0 - 1 + row, gw = w0 - 1 + col;
            const bool ok = v < MVOX && (unsigned)gh < (unsigned)H && (unsigned)gw < (unsigned)W;
            const unsigned vr = (unsigned)(row * W + col);
            vof[j] = fastdma ? (ok ? vr * (unsigned)(q.in_ld * (int)sizeof(T)) + pieceb : kDmaOob) : vr;
            okmask |= (ok ? 1u : 0u) << j;
        }
        const int pch = PE * (lane & 1);
        const unsigned lds_mine = __builtin_amdgcn_readfirstlane((unsigned)(size_t)mybuf);

        struct Dma { i32x4 rs; unsigned ldb, dst; bool on, chok; };
        auto dma_open = [&](int it) -> Dma {   // item it = (plane plo + it / nchunks, chunk it % nchunks)
            Dma d;
            d.on = it < nitems;
            const int p = plo + it / q.nchunks, c = it % q.nchunks;
            const bool second = x2 != nullptr && c * CK >= q.ksplit;   // wave-uniform: which tensor holds this chunk
            const T* xs = second ? x2 : x;
            const int ld = second ? q.in2_ld : q.in_ld, c0 = second ? c * CK - q.ksplit : c * CK;
            d.ldb = (unsigned)ld * (unsigned)sizeof(T);
            const unsigned long long org =
                (unsigned long long)(xs + (((((int64_t)n * D + p) * H + (h0 - 1)) * W + (w0 - 1)) * ld + c0));
            d.rs[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)(org & 0xffffffffu));
            d.rs[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)((org >> 32) & 0xffffu));
            d.rs[2] = (int)kDmaRecords;
            d.rs[3] = 0x00020000;
            d.dst = lds_mine + (unsigned)(it % q.nbuf) * (unsigned)MBUF;
            d.chok = c * CK + pch < q.Kc;   // bf16 with Kc % 16 == 8: the last chunk's upper piece is zero-filled
            return d;
        };
        auto dma_piece = [&](const Dma& d, int j) {
            if (!d.on) return;   // wave-uniform
            // (the sixth piece covers image bytes 5120 .. 5759: lanes 40..63 would write into the next buffer)
            if (j < MPIECES - 1 || lane < (MBUF - (MPIECES - 1) * 1024) / 16) {
                if (fastdma) {
                    lds_dma16(vof[j], d.rs, d.dst + (unsigned)j * 1024u);
                } else {
                    const bool ok = ((okmask >> j) & 1u) && d.chok;
                    lds_dma16(ok ? __umul24(vof[j], d.ldb) + pieceb : kDmaOob, d.rs, d.dst + (unsigned)j * 1024u);
                }
            }
        };

        // fragment byte offsets inside a plane buffer (row i adds i * MROW)
        const int offA = (li + (kq >> 1)) * 32 + 16 * (kq & 1);                  // paired (kw = 0 | kw = 1), halo rows 0..9
        const int offB = ((kq >> 1) * MHW + li + 2) * 32 + 16 * (kq & 1);        // paired (kh = 0 | kh = 1) at kw = 2, rows m, m+1
        const int offC = (2 * MHW + li + 2) * 32 + 8 * kq;                       // single tap (2, 2): row m + 2, 8 bytes per lane
        const unsigned char* const wl = smem + lane * 16;
        const unsigned char* const wl8 = smem + 4096 + lane * 8;

        // Output plane o lives in accumulator slot o mod 3 — FIXED registers for the whole march.  What rotates with the input plane
        // p is which tap plane each slot receives: slot s takes kd(s) with (kd(0), kd(1), kd(2)) a cyclic shift of (1, 0, 2) that
        // advances by one per plane, i.e. only the LDS address of the slot's weight fragments changes (three scalars).  One body,
        // no accumulator ever moves or is cleared: the first MFMA into a re-used slot (its kd = 0 group of the plane's first chunk)
        // takes C = 0.  Every tap is multiplied for every input plane of the segment; output planes outside [dlo, dhi) are simply
        // never stored — (L + 2) / L of the minimal MFMA work for a segment of L planes.
        // accumulator (slot s, row m) = v[160 + 4*(8*s + m) .. + 3] (see the helpers above): declared, then cleared once
        asm volatile("" ::: MRI3D_ACC_CLOBBERS);
        static_for<96>([&](auto i) { asm volatile("v_mov_b32 v%c0, 0" ::"n"(ACC0 + decltype(i)::value)); });
        acc_fence();

        // the lane's accumulator rows are channels 4*sigma(kq) .. + 3 of the N-block (pack_w_march_kernel)
        const int sg = (kq & 1) * 2 + (kq >> 1);
        const int co = nb * 16 + 4 * sg;
        // (the lane's bias quad is read from LDS where a plane is stored: four VGPRs less across the march)
        // where this N-block's outputs go (data gradient over a split operand: the second tensor)
        const bool osecond = y2 != nullptr && nb * 16 >= q.nsplit;
        T* const yd = osecond ? y2 : y;
        const int yld = osecond ? q.out2_ld : q.out_ld, cbase = osecond ? nb * 16 - q.nsplit : nb * 16;
        const unsigned row_step = (unsigned)(W * yld);
        // fp32: a lane stores its own quad (16 bytes) of every row.  bf16: lanes 0-31 (kq = 0, 1) store channels 8*kq .. + 7 of the
        // EVEN rows, lanes 32-63 those of the ODD rows (their partner's quad arrives by v_permlane32_swap): 16-byte stores, half as many
        const unsigned lane_off = kBf16 ? (unsigned)(li * yld + 8 * (kq & 1)) + (kq >> 1) * row_step : (unsigned)(li * yld + 4 * sg);
        const bool lane_ok = (kBf16 ? nb * 16 + 8 * (kq & 1) : co) < q.Nc && w0 + li < W;
        const int row_par = kBf16 ? (kq >> 1) : 0;   // the row of a pair this lane stores
        const int nstores = (h0 + MH <= H) ? (kBf16 ? MH / 2 : MH) : 0;   // store instructions per plane that certainly issue (wave-uniform)
        // BatchNorm batch statistics of a = y - bias: per-lane fp32 sums over the column's planes (at most 8 rows x seglen values
        // per lane), reduced over the 16 voxel lanes and handed on in float64 once, at the end of the march
        f32x2 s1a = f32x2{0.f, 0.f}, s1b = f32x2{0.f, 0.f}, s2a = f32x2{0.f, 0.f}, s2b = f32x2{0.f, 0.f};

        // store output plane o from accumulator slot S (compile-time).  Reads only: the slot is re-used without being cleared.
        auto store_slot = [&](auto slotc, int o) __attribute__((always_inline)) {
            constexpr int S = decltype(slotc)::value;
            acc_fence();   // the last MFMA into this slot may have been issued just above
            T* const ytile = yd + (((((int64_t)n * D + o) * H + h0) * W + w0) * yld + cbase);
            f32x2 b01 = f32x2{0.f, 0.f}, b23 = f32x2{0.f, 0.f};
            if constexpr (BIAS) {
                const f32x4 bq = *reinterpret_cast<const f32x4*>(bias_lds + 4 * sg);
                b01 = f32x2{bq[0], bq[1]}, b23 = f32x2{bq[2], bq[3]};
            }
            // two rows at a time, straight from the named registers: statistics, bias, conversion, lane exchange, one store
            static_for<MH / 2>([&](auto pc) {
                constexpr int m = 2 * decltype(pc)::value, A0 = 4 * (8 * S + m), A1 = A0 + 4;
                if constexpr (STATS) {
                    if (h0 + m < H) {   // wave-uniform
                        acc_stat<A0>(s1a, s2a);
                        acc_stat<A0 + 2>(s1b, s2b);
                    }
                    if (h0 + m + 1 < H) {
                        acc_stat<A1>(s1a, s2a);
                        acc_stat<A1 + 2>(s1b, s2b);
                    }
                }
                if constexpr (kBf16) {   // (the plan guarantees Nc % 8 == 0 and a pitch of whole octets)
                    unsigned p00, p01, p10, p11;
                    if constexpr (BIAS) {
                        p00 = __builtin_bit_cast(unsigned, __builtin_convertvector(acc_pk_add<A0>(b01), bf16x2_t));
                        p01 = __builtin_bit_cast(unsigned, __builtin_convertvector(acc_pk_add<A0 + 2>(b23), bf16x2_t));
                        p10 = __builtin_bit_cast(unsigned, __builtin_convertvector(acc_pk_add<A1>(b01), bf16x2_t));
                        p11 = __builtin_bit_cast(unsigned, __builtin_convertvector(acc_pk_add<A1 + 2>(b23), bf16x2_t));
                    } else {
                        p00 = acc_cvt_pk_bf16<A0>(), p01 = acc_cvt_pk_bf16<A0 + 2>(), p10 = acc_cvt_pk_bf16<A1>(), p11 = acc_cvt_pk_bf16<A1 + 2>();
                    }
                    // half exchange: lanes 0-31 end with (own row m, partner's row m), lanes 32-63 with (partner's row m+1, own row m+1)
                    auto rx = __builtin_amdgcn_permlane32_swap(p00, p10, false, false);
                    auto ry = __builtin_amdgcn_permlane32_swap(p01, p11, false, false);
                    if (lane_ok && h0 + m + row_par < H)
                        *reinterpret_cast<uint4*>(ytile + (lane_off + (unsigned)m * row_step)) = make_uint4(rx[0], ry[0], rx[1], ry[1]);
                } else if (lane_ok) {   // the plan guarantees Nc % 4 == 0 and a pitch of whole quads: one vector store per row
                    if constexpr (BIAS) {
                        if (h0 + m < H) {   // wave-uniform
                            const f32x2 lo = acc_pk_add<A0>(b01), hi = acc_pk_add<A0 + 2>(b23);
                            stf4(ytile + (lane_off + (unsigned)m * row_step), make_float4(lo[0], lo[1], hi[0], hi[1]));
                        }
                        if (h0 + m + 1 < H) {
                            const f32x2 lo = acc_pk_add<A1>(b01), hi = acc_pk_add<A1 + 2>(b23);
                            stf4(ytile + (lane_off + (unsigned)(m + 1) * row_step), make_float4(lo[0], lo[1], hi[0], hi[1]));
                        }
                    } else {
                        if (h0 + m < H) acc_store4<A0>(ytile + (lane_off + (unsigned)m * row_step));
                        if (h0 + m + 1 < H) acc_store4<A1>(ytile + (lane_off + (unsigned)(m + 1) * row_step));
                    }
                }
            });
        };
        auto store_plane = [&](int o) __attribute__((always_inline)) {   // slot o mod 3 (o >= 0)
            const int s = o % 3;
            if (s == 0) store_slot(std::integral_constant<int, 0>{}, o);
            else if (s == 1) store_slot(std::integral_constant<int, 1>{}, o);
            else store_slot(std::integral_constant<int, 2>{}, o);
        };

#if defined(MRI3D_EXPERIMENT_STAMPS)
        unsigned long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const unsigned long long clk0 = __builtin_amdgcn_s_memtime(), ref0 = __builtin_amdgcn_s_memrealtime();
#endif
        // prologue: the first PD items' pieces
        for (int k = 0; k < PD; ++k) {
            const Dma d0 = dma_open(k);
#pragma unroll
            for (int j = 0; j < MPIECES; ++j) dma_piece(d0, j);
        }
        // kd of slot s while plane p is consumed: output plane o = p + 1 - kd must have o mod 3 == s
        int kds[3];
#pragma unroll
        for (int s = 0; s < 3; ++s) kds[s] = 2 - ((s - plo + 1) % 3 + 3) % 3;
        // ONE flat loop over the items (plane p, chunk c), not peeled or unrolled: with a chunk loop nested in a plane loop hipcc
        // kept the accumulators in different registers inside and outside the inner loop (~100 moves per plane)
        int p = plo, c = 0;
#pragma clang loop unroll(disable)
        for (int it = 0; it < nitems; ++it) {
            {
                MARCH_STAMP(t_top);
                // a new input plane: output plane p - 2 was completed by the previous one (its slot is the one whose kd is 0 now)
                const bool stored = c == 0 && p - 2 >= dlo;   // wave-uniform
                if (stored) store_plane(p - 2);
                MARCH_STAMP(t_c0);
                MARCH_STAMP_ADD(0, t_top, t_c0);
                // The DMA pieces of this item have landed once only the younger vector-memory operations are outstanding (they
                // return in order): the pieces of item it + 1 when the lead is two items, and the stores just issued.  (A column
                // that is ragged in h issues fewer stores than rows: its stores are not counted, i.e. waited for as well.)
                {
                    const int younger = ((PD == 2 && it + 1 < nitems) ? 6 : 0) + (stored ? nstores : 0);
                    if (younger == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    else if (younger == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                    else if (younger == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                    else if (younger == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                    else if (younger == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
                }
                MARCH_STAMP(t_wait);
                MARCH_STAMP_ADD(1, t_c0, t_wait);
                const Dma dn = dma_open(it + PD);
                const unsigned char* const buf = mybuf + (it % q.nbuf) * MBUF;
                // the three slots' weight fragments of this chunk: [kd(s)][group][lane] — scalar byte offsets, added to the lane's
                // address at each read (three more VGPR pairs held across the item cost more than the adds)
                int wso[3];
#pragma unroll
                for (int s = 0; s < 3; ++s) wso[s] = c * MWCHUNK + kds[s] * MWKD;
                auto ws16 = [&](int s) -> const unsigned char* { return wl + wso[s]; };
                auto ws8 = [&](int s) -> const unsigned char* { return wl8 + wso[s]; };

                // Phase A: per slot the three kh groups on the ten row fragments of the (kw = 0 | kw = 1) pairing — a group is
                // 8 MFMAs, one per output row, on one weight fragment (3-deep ring, two groups of lead)
                f32x4 FA[MHH];
#pragma unroll
                for (int i = 0; i < MHH; ++i) FA[i] = *reinterpret_cast<const f32x4*>(buf + offA + i * MROW);
                f32x4 wq[3];
                wq[0] = *reinterpret_cast<const f32x4*>(ws16(0));
                wq[1] = *reinterpret_cast<const f32x4*>(ws16(0) + 1024);
                f32x4 wB[3], FB[3];   // phase B: the kw = 2 pairing (kh = 0 | kh = 1) of the three slots, row fragments streamed two rows ahead
                f32x2 wC[3], FC[3];   // phase C: the single tap (kh = 2, kw = 2) of the three slots on the K = 16 MFMA
                static_for<9>([&](auto seqc) {
                    constexpr int seq = decltype(seqc)::value, sl = seq / 3, kh = seq % 3;
                    if constexpr (seq + 2 < 9) wq[(seq + 2) % 3] = *reinterpret_cast<const f32x4*>(ws16((seq + 2) / 3) + ((seq + 2) % 3) * 1024);
                    if constexpr (seq < MPIECES) dma_piece(dn, seq);
                    if constexpr (seq >= 6) wB[seq - 6] = *reinterpret_cast<const f32x4*>(ws16(seq - 6) + 3 * 1024);
                    if constexpr (seq >= 7) FB[seq - 7] = *reinterpret_cast<const f32x4*>(buf + offB + (seq - 7) * MROW);
                    __builtin_amdgcn_sched_barrier(0);   // keep the prefetches above this group's MFMAs
                    const f32x4 wf = wq[seq % 3];
                    if (kh == 0 && kds[sl] == 0 && c == 0) {   // (wave-uniform) the slot is re-used: start from zero
                        if constexpr (kBf16) {
                            static_for<MH>([&](auto mc) { constexpr int m = decltype(mc)::value; acc_bf16_k32_first<4 * (8 * sl + m)>(wf, FA[m]); });
                        } else {
                            static_for<MH>([&](auto mc) { constexpr int m = decltype(mc)::value; acc_f32_k4_first<4 * (8 * sl + m)>(wf[0], FA[m][0]); });
                            static_for<3>([&](auto sc) {
                                static_for<MH>([&](auto mc) {
                                    constexpr int m = decltype(mc)::value, s4 = decltype(sc)::value + 1;
                                    acc_f32_k4<4 * (8 * sl + m)>(wf[s4], FA[m][s4]);
                                });
                            });
                        }
                    } else {
                        if constexpr (kBf16) {
                            static_for<MH>([&](auto mc) { constexpr int m = decltype(mc)::value; acc_bf16_k32<4 * (8 * sl + m)>(wf, FA[m + kh]); });
                        } else {
                            static_for<4>([&](auto sc) {
                                static_for<MH>([&](auto mc) {
                                    constexpr int m = decltype(mc)::value, s4 = decltype(sc)::value;
                                    acc_f32_k4<4 * (8 * sl + m)>(wf[s4], FA[m + kh][s4]);
                                });
                            });
                        }
                    }
                });
                // Phase B: row m of all three output planes from one row fragment
                static_for<MH>([&](auto mc) {
                    constexpr int m = decltype(mc)::value;
                    if constexpr (m + 2 < MH) FB[(m + 2) % 3] = *reinterpret_cast<const f32x4*>(buf + offB + (m + 2) * MROW);
                    if constexpr (m >= MH - 3) wC[m - (MH - 3)] = *reinterpret_cast<const f32x2*>(ws8(m - (MH - 3)));
                    if constexpr (m >= MH - 2) FC[m - (MH - 2)] = *reinterpret_cast<const f32x2*>(buf + offC + (m - (MH - 2)) * MROW);
                    __builtin_amdgcn_sched_barrier(0);
                    const f32x4 f = FB[m % 3];
                    if constexpr (kBf16) {
                        static_for<3>([&](auto sc) { constexpr int sl = decltype(sc)::value; acc_bf16_k32<4 * (8 * sl + m)>(wB[sl], f); });
                    } else {
                        static_for<4>([&](auto s4c) {
                            static_for<3>([&](auto sc) {
                                constexpr int sl = decltype(sc)::value, s4 = decltype(s4c)::value;
                                acc_f32_k4<4 * (8 * sl + m)>(wB[sl][s4], f[s4]);
                            });
                        });
                    }
                });
                // Phase C
                static_for<MH>([&](auto mc) {
                    constexpr int m = decltype(mc)::value;
                    if constexpr (m + 2 < MH) FC[(m + 2) % 3] = *reinterpret_cast<const f32x2*>(buf + offC + (m + 2) * MROW);
                    __builtin_amdgcn_sched_barrier(0);
                    const f32x2 f = FC[m % 3];
                    if constexpr (kBf16) {
                        static_for<3>([&](auto sc) { constexpr int sl = decltype(sc)::value; acc_bf16_k16<4 * (8 * sl + m)>(wC[sl], f); });
                    } else {
                        static_for<2>([&](auto s2c) {
                            static_for<3>([&](auto sc) {
                                constexpr int sl = decltype(sc)::value, s2 = decltype(s2c)::value;
                                acc_f32_k4<4 * (8 * sl + m)>(wC[sl][s2], f[s2]);
                            });
                        });
                    }
                });
                MARCH_STAMP(t_end);
                MARCH_STAMP_ADD(2, t_wait, t_end);
                MARCH_STAMP_ADD(3, 0ull, 1ull);
            }
            if (++c == q.nchunks) {
                c = 0;
                ++p;
#pragma unroll
                for (int s = 0; s < 3; ++s) kds[s] = kds[s] == 2 ? 0 : kds[s] + 1;
            }
        }
        // input plane phi was the last: it completed output plane phi - 1; plane phi itself is complete when the volume ends there
        if (phi - 1 >= dlo) store_plane(phi - 1);
        if (phi < dhi) store_plane(phi);
#if defined(MRI3D_EXPERIMENT_STAMPS)
        if (blockIdx.x == 0 && tid == 0) {
            stamp_acc[6] = __builtin_amdgcn_s_memtime() - clk0;
            stamp_acc[7] = __builtin_amdgcn_s_memrealtime() - ref0;
            for (int i = 0; i < 8; ++i) g_march_stamps[i] += stamp_acc[i];
        }
        if (tid == 0 && blockIdx.x < 1024) {
            g_march_span[4 * blockIdx.x] = span_t0;
            g_march_span[4 * blockIdx.x + 1] = ref0;
            g_march_span[4 * blockIdx.x + 2] = __builtin_amdgcn_s_memrealtime();
        }
#endif
        if constexpr (STATS) {   // the lane's sums -> the 16 voxel lanes' totals -> this wave's float64 slot
            const bool vok = w0 + li < W;
            float s1[4] = {s1a[0], s1a[1], s1b[0], s1b[1]}, s2[4] = {s2a[0], s2a[1], s2b[0], s2b[1]};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                s1[r] = row_sum16(vok ? s1[r] : 0.f);
                s2[r] = row_sum16(vok ? s2[r] : 0.f);
            }
            if (li == 0) {
                double* slot = stat_lds + ((size_t)wv * 16 + 4 * sg) * 2;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    slot[2 * r] = (double)s1[r];
                    slot[2 * r + 1] = (double)s2[r];
                }
            }
        }
    }

    if constexpr (STATS) {
        __syncthreads();   // every wave of the workgroup arrives here (inactive ones went straight to it)
        for (int i = tid; i < q.Nc * 2; i += WV * 64) {
            const int chn = i >> 1, st = i & 1, rel = chn - nb * 16;
            double v = 0.0;
            if (rel >= 0 && rel < 16) {
#pragma unroll
                for (int w8 = 0; w8 < MWAVES; ++w8) v += stat_lds[((size_t)w8 * 16 + rel) * 2 + st];
            }
            stat_part[(size_t)blockIdx.x * q.Nc * 2 + i] = v;
        }
    }
}

template <typename T, bool STATS, bool BIAS>
__global__ void __launch_bounds__(512, 2) __attribute__((amdgpu_num_vgpr(160)))   // hipcc's own values: v0..v159; v160..v255 = accumulators
conv_march_kernel(const T* __restrict__ x, const T* __restrict__ x2, const unsigned char* __restrict__ wp,
                  const float* __restrict__ bias, T* __restrict__ y, T* __restrict__ y2, double* __restrict__ stat_part,
                  const MarchGeom q) {
    conv_march_body<T, STATS, BIAS, 8>(x, x2, wp, bias, y, y2, stat_part, q);
}

// ------------------------------------------------------------------------------------------------ host side
struct MarchPlan {
    MarchGeom q;
    int grid;
    size_t smem, wp_bytes;
};

// Does the marching kernel take this layer, and how?  `force`: the explicit entry points of the parity tests take every
// geometry the kernel can compute; the dispatcher (force = false) only takes layers where it is the faster kernel: bf16
// tensors, a grid that fills the chip, segments long enough to amortise their two halo planes.
bool conv_march_plan(const Mri3dConvGeom& g, bool dgrad, bool stats, bool force, MarchPlan& p) {
    if (!(g.kd == 3 && g.kh == 3 && g.kw == 3 && g.sd == 1 && g.sh == 1 && g.sw == 1 && g.pd == 1 && g.ph == 1 && g.pw == 1 &&
          g.dd == 1 && g.dh == 1 && g.dw == 1))
        return false;
    const bool bf = g.dtype == MRI3D_BF16;
    const int Kc = dgrad ? g.co : g.ci, Nc = dgrad ? g.ci : g.co;
    const int in_ld = dgrad ? g.y_ld : g.x_ld, out_ld = dgrad ? g.x_ld : g.y_ld;
    if (Kc % 8 != 0 || in_ld % (bf ? 8 : 4) != 0 || Nc < 8) return false;
    if (Nc % 4 != 0 || out_ld % 4 != 0) return false;     // the epilogue stores whole channel quads
    if (bf && (Nc % 8 != 0 || out_ld % 8 != 0)) return false;   // ... bf16: whole octets (16-byte stores after the lane exchange)
    MarchGeom& q = p.q;
    q.N = g.n, q.D = g.di, q.H = g.hi, q.W = g.wi, q.Kc = Kc, q.in_ld = in_ld, q.Nc = Nc, q.out_ld = out_ld;
    const int CK = bf ? 16 : 8;
    q.nchunks = cdiv(Kc, CK);
    q.NTT = cdiv(Nc, 16);
    if (q.nchunks > 4) return false;                       // weights of one N-block + the plane buffers must fit 160 KiB of LDS
    if (stats && q.NTT > 8) return false;
    if ((int64_t)g.n * g.di * g.hi * g.wi * std::max(in_ld, out_ld) > ((int64_t)1 << 40)) return false;
    const int colsH = cdiv(g.hi, MH), colsW = cdiv(g.wi, MW);
    q.waves = 8;   // (twelve waves — three per SIMD — were built and measured: the item body has to shrink to 72 VGPRs for it, and the
                   //  streamed fragments that takes cost more than the third wave brings: 16 -> 16 0.158 against 0.150 ms)
    q.nbuf = q.nchunks <= 1 ? 3 : 2;
    // shape of a workgroup's columns (wgh x wgw = waves): least padding first, then the squarest
    int besth = -1, bestpad = 0, bestsq = 0;
    for (int gh = 1; gh <= q.waves; ++gh) {
        if (q.waves % gh) continue;
        const int gw = q.waves / gh;
        const int pad = cdiv(colsH, gh) * gh * cdiv(colsW, gw) * gw;
        const int sq = -std::abs(gh * MH - gw * MW);   // squarest footprint
        if (besth < 0 || pad < bestpad || (pad == bestpad && sq > bestsq)) besth = gh, bestpad = pad, bestsq = sq;
    }
    q.wgh = besth, q.wgw = q.waves / besth;
    q.gch = cdiv(colsH, q.wgh), q.gcw = cdiv(colsW, q.wgw);
    const int64_t wgcols = (int64_t)q.gch * q.gcw * g.n * q.NTT;
    // d segments: fill the 256 CUs in whole rounds, keep a segment's two halo planes small against its length
    int bestS = 1;
    double beste = -1.0;
    const int maxS = std::max(1, g.di / 8);
    for (int S = 1; S <= maxS; ++S) {
        const int len = cdiv(g.di, S);
        if (cdiv(g.di, len) != S) continue;   // S segments of len planes, the last one possibly shorter
        const int64_t units = wgcols * S;
        const double fill = (double)units / (double)(cdiv64(units, 256) * 256);
        const double e = fill * len / (len + 2.0);
        if (e > beste + 1e-9) beste = e, bestS = S;
    }
    q.nseg = bestS;
    q.seglen = cdiv(g.di, bestS);
    q.ksplit = q.in2_ld = q.nsplit = q.out2_ld = 0;
    q.stagger = tuning_knob("MRI3D_MARCH_STAGGER", 40);
    const int64_t grid = wgcols * q.nseg;
    if (grid > 0x7fffffff) return false;
    p.grid = (int)grid;
    p.wp_bytes = (size_t)q.NTT * q.nchunks * MWCHUNK;
    p.smem = (size_t)((q.nchunks * MWCHUNK + 1023) & ~1023) + (size_t)q.waves * q.nbuf * MBUF + (size_t)q.waves * 32 * sizeof(double) + 64;   // statistics slots + the bias block
    if (p.smem > 160 * 1024) return false;
    if (force) return true;
    // the dispatcher's choice (measured on MI355X, profiles/r03_*): bf16 tensors; at least 3/4 of a round of workgroups;
    // efficiency of the (fill, halo) split at least 0.7
#if defined(MRI3D_NO_MARCH)   // tuning builds (tools/march_bench.py --lib): the round-2 dispatcher, tiled kernel everywhere
    return false;
#endif
    // fp32: the tiled kernel is MFMA-bound like this one; the march is 2-3 % ahead on the full-resolution layers with 16 output
    // channels per pass (16 -> 16 128.5 against 126.3, 16 -> 48 data gradient 130.1 against 127.4 TFLOP/s) and behind elsewhere
    if (!bf && (Nc % 16 != 0 || (int64_t)g.n * g.di * g.hi * g.wi < (int64_t)4 << 20)) return false;
    if (grid < 192 || beste < 0.70) return false;
    // one 16-channel output block per pass: wider outputs re-stage the input once per block, which only pays for the data
    // gradient of the decoder's 16 -> 48 layer (the tiled kernel makes three passes there too)
    if (!(q.NTT == 1 || (dgrad && q.NTT <= 3))) return false;
    return true;
}

bool conv_march_takes(const Mri3dConvGeom& g, bool dgrad, bool stats, bool force) {
    MarchPlan p;
    return conv_march_plan(g, dgrad, stats, force, p);
}

size_t conv_march_workspace_bytes(const Mri3dConvGeom& g, bool dgrad) {
    MarchPlan p;
    return conv_march_plan(g, dgrad, false, true, p) ? p.wp_bytes : 0;
}

int conv_march_stat_blocks(const Mri3dConvGeom& g, bool force) {
    MarchPlan p;
    return conv_march_plan(g, false, true, force, p) ? p.grid : 0;
}

// second tensor of a split operand: forward, input channels >= split live in `second`; data gradient, output channels >= split
int conv_march_run(const Mri3dConvGeom& g, bool dgrad, bool force, const void* in_v, const float* w, const float* bias, void* out_v,
                   void* ws, size_t ws_bytes, hipStream_t s, double* stat_part, const void* second, int split, int second_ld) {
    MarchPlan p;
    MRI3D_REQUIRE(conv_march_plan(g, dgrad, stat_part != nullptr, force, p), MRI3D_ENOTSUP, "conv3d(march): unsupported geometry");
    MRI3D_REQUIRE(ws && ws_bytes >= p.wp_bytes, MRI3D_EWORKSPACE, "conv3d(march): workspace %zu < %zu", ws_bytes, p.wp_bytes);
    MRI3D_REQUIRE(aligned16(in_v, out_v, ws) && aligned16(second), MRI3D_EINVAL, "conv3d(march): tensors / workspace must be 16-byte aligned");
    MarchGeom q = p.q;
    const bool bf = g.dtype == MRI3D_BF16;
    if (second) {
        MRI3D_REQUIRE(split > 0 && split % 16 == 0 && second_ld % (bf ? 8 : 4) == 0 && second_ld % 4 == 0, MRI3D_ENOTSUP,
                      "conv3d(march): split must be a multiple of 16");
        MRI3D_REQUIRE(!(bf && dgrad) || second_ld % 8 == 0, MRI3D_ENOTSUP, "conv3d(march): bf16 output pitch must be a multiple of 8");
        if (dgrad) q.nsplit = split, q.out2_ld = second_ld;
        else {
            MRI3D_REQUIRE(split % (bf ? 16 : 8) == 0, MRI3D_ENOTSUP, "conv3d(march): split inside a chunk");
            q.ksplit = split, q.in2_ld = second_ld;
        }
    }
    const int total_el = (int)(p.wp_bytes / (bf ? 2 : 4));
    if (bf)
        hipLaunchKernelGGL(pack_w_march_kernel<bf16_t>, dim3(std::min(cdiv(total_el, 256), 2048)), dim3(256), 0, s, w,
                           static_cast<bf16_t*>(ws), g.co, g.ci, dgrad ? 1 : 0, q.nchunks, q.NTT);
    else
        hipLaunchKernelGGL(pack_w_march_kernel<float>, dim3(std::min(cdiv(total_el, 256), 2048)), dim3(256), 0, s, w,
                           static_cast<float*>(ws), g.co, g.ci, dgrad ? 1 : 0, q.nchunks, q.NTT);
    const void* x2 = dgrad ? nullptr : second;
    void* y2 = dgrad ? const_cast<void*>(second) : nullptr;
#define MRI3D_MARCH_CASE(STv, BIv)                                                                                      \
    if ((stat_part != nullptr) == STv && (bias != nullptr) == BIv) {                                                    \
        auto k8 = conv_march_kernel<T, STv, BIv>;                                                                       \
        static const hipError_t attr8 = hipFuncSetAttribute(reinterpret_cast<const void*>(k8),                          \
                                                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);   \
        (void)attr8;   /* once per kernel, not per launch */                                                            \
        hipLaunchKernelGGL(k8, dim3(p.grid), dim3(512), p.smem, s, (const T*)in_v, (const T*)x2,                        \
                           (const unsigned char*)ws, bias, (T*)out_v, (T*)y2, stat_part, q);                            \
    }
    MRI3D_DISPATCH_DTYPE(g.dtype, T, {
        MRI3D_MARCH_CASE(false, false)
        MRI3D_MARCH_CASE(false, true)
        MRI3D_MARCH_CASE(true, false)
        MRI3D_MARCH_CASE(true, true)
    });
#undef MRI3D_MARCH_CASE
    return check_launch(dgrad ? "conv3d_dgrad(march)" : "conv3d_fwd(march)");
}

}  // namespace mri3d
