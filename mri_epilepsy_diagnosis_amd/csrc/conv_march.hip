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

namespace mri3d {

constexpr int MH = 8, MW = 16;              // a wave's output column: rows x voxels
constexpr int MHH = MH + 2, MHW = MW + 2;   // halo plane 10 x 18
constexpr int MVOX = MHH * MHW;             // 180 voxels
constexpr int MPIECES = 6;                  // 1-KiB DMA wave-instructions per plane chunk (360 of 384 16-byte pieces used)
constexpr int MBUF = MPIECES * 1024;        // bytes per LDS plane buffer
constexpr int MWKD = 4 * 1024 + 512;        // packed weight bytes per (chunk, kd): four paired groups + the single tap
constexpr int MWCHUNK = 3 * MWKD;           // 13,824
constexpr int MWAVES = 8;                   // waves (columns) per workgroup
constexpr int MROW = MHW * 32;              // LDS bytes per halo row (576)

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// Packed weights of the marching kernel: image [N-block nb][chunk][kd][group g][lane][elements]
//   g = 0..2 (paired, kh = g):  k-slot ks = lane >> 5 carries tap (kd, kh, kw = ks);  g = 3 (paired): tap (kd, kh = ks, kw = 2)
//       element s (PE per lane: 8 bf16 / 4 fp32) = channel chunk*CK + PE*((lane >> 4) & 1) + s
//   g = 4 (single tap (kd, 2, 2)): element s (PE/2 per lane) = channel chunk*CK + (PE/2)*(lane >> 4) + s
//   output channel nc = nb*16 + (lane & 15);  forward: W[nc][kc][tap];  data gradient: W[kc][nc][26 - tap]
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
        const int nc = nb * 16 + (lane & 15), tap = (kd * 3 + kh) * 3 + kw;
        float v = 0.f;
        if (nc < Nc && kc < Kc) v = dgrad ? w[((size_t)kc * Ci + nc) * 27 + (26 - tap)] : w[((size_t)nc * Ci + kc) * 27 + tap];
        wp[i] = (WT)v;
    }
}

struct MarchGeom {
    int N, D, H, W, Kc, in_ld, Nc, out_ld;
    int nchunks, NTT;      // 32-byte input chunks; 16-channel output blocks (one per workgroup)
    int wgh, wgw;          // the 8 wave-columns of a workgroup form wgh x wgw columns (h x w)
    int gch, gcw;          // workgroup-columns along h and w
    int nseg, seglen;      // d segments of seglen planes
    int nbuf;              // LDS plane buffers per wave (2 or 3)
    int ksplit, in2_ld;    // forward over cat((x, x2)): input channels >= ksplit live in x2 (pitch in2_ld); 0: one tensor
    int nsplit, out2_ld;   // its data gradient: output channels >= nsplit go to y2; 0: one tensor
};

template <typename T, bool STATS>
__global__ void __launch_bounds__(512, 2)
conv_march_kernel(const T* __restrict__ x, const T* __restrict__ x2, const unsigned char* __restrict__ wp,
                  const float* __restrict__ bias, T* __restrict__ y, T* __restrict__ y2, double* __restrict__ stat_part,
                  const MarchGeom q) {
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
        for (int i = tid * 16; i < wbytes; i += 512 * 16) *reinterpret_cast<uint4*>(smem + i) = *reinterpret_cast<const uint4*>(wsrc + i);
    }
    const int bufbase = (wbytes + 1023) & ~1023;
    unsigned char* const mybuf = smem + bufbase + wv * q.nbuf * MBUF;
    double* const stat_lds = reinterpret_cast<double*>(smem + bufbase + MWAVES * q.nbuf * MBUF);   // [8 waves][16 channels][2]
    if constexpr (STATS) {
        if (tid < MWAVES * 32) stat_lds[tid] = 0.0;
    }
    __syncthreads();

    const int wy = wv / q.wgw, wx = wv - wy * q.wgw;
    const int h0 = (ch * q.wgh + wy) * MH, w0 = (cw * q.wgw + wx) * MW;
    const int dlo = seg * q.seglen, dhi = min(dlo + q.seglen, q.D);
    const bool active = h0 < q.H && w0 < q.W && dlo < q.D;

    if (active) {
        const int D = q.D, H = q.H, W = q.W;
        const int plo = max(dlo - 1, 0), phi = min(dhi, D - 1);   // input planes this column consumes
        const int nitems = (phi - plo + 1) * q.nchunks;
        const int PD = q.nbuf - 1;                                // items of DMA lead

        // per-lane DMA geometry, fixed for the column: piece j of the lane = 16-byte piece j*64 + lane of the [voxel][32 B] image
        unsigned vrel[MPIECES], okmask = 0;
#pragma unroll
        for (int j = 0; j < MPIECES; ++j) {
            const int v = (j * 64 + lane) >> 1;
            const int row = v / MHW, col = v - row * MHW;
            const int gh = h0 - 1 + row, gw = w0 - 1 + col;
            const bool ok = v < MVOX && (unsigned)gh < (unsigned)H && (unsigned)gw < (unsigned)W;
            vrel[j] = (unsigned)(row * W + col);
            okmask |= (ok ? 1u : 0u) << j;
        }
        const unsigned pieceb = (unsigned)(lane & 1) * 16u;
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
            const bool ok = ((okmask >> j) & 1u) && d.chok;
            lds_dma16(ok ? __umul24(vrel[j], d.ldb) + pieceb : kDmaOob, d.rs, d.dst + (unsigned)j * 1024u);
        };

        // fragment byte offsets inside a plane buffer (row i adds i * MROW)
        const int offA = (li + (kq >> 1)) * 32 + 16 * (kq & 1);                  // paired (kw = 0 | kw = 1), halo rows 0..9
        const int offB = ((kq >> 1) * MHW + li + 2) * 32 + 16 * (kq & 1);        // paired (kh = 0 | kh = 1) at kw = 2, rows m, m+1
        const int offC = (2 * MHW + li + 2) * 32 + 8 * kq;                       // single tap (2, 2): row m + 2, 8 bytes per lane
        const unsigned char* const wl = smem + lane * 16;
        const unsigned char* const wl8 = smem + 4096 + lane * 8;

        // acc[0]: output plane p - 1 (kd = 2), acc[1]: plane p (kd = 1), acc[2]: plane p + 1 (kd = 0) while input plane p is consumed.
        // The roles are fixed and the registers ROTATE when a plane is done (acc[0] stored, acc[0] <- acc[1] <- acc[2] <- 0): the body is
        // one straight-line block.  (Compile-time slots o % 3 selected by a switch on p % 3 made hipcc keep two copies of the 96
        // accumulator registers and spill.)  Every tap is multiplied for every input plane of the segment: planes whose output lies
        // outside [dlo, dhi) just rotate out unstored — (L + 2) / L of the minimal MFMA work for a segment of L planes.
        f32x4 acc[3][MH];
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int m = 0; m < MH; ++m) acc[s][m] = f32x4{0.f, 0.f, 0.f, 0.f};

        // bias quad of the lane (channels nb*16 + 4*kq ..)
        const int co = nb * 16 + 4 * kq;
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (bias && co < q.Nc) bv = make_float4(bias[co], bias[co + 1], bias[co + 2], bias[co + 3]);   // Nc % 4 == 0
        // where this N-block's outputs go (data gradient over a split operand: the second tensor)
        const bool osecond = y2 != nullptr && nb * 16 >= q.nsplit;
        T* const yd = osecond ? y2 : y;
        const int yld = osecond ? q.out2_ld : q.out_ld, cbase = osecond ? nb * 16 - q.nsplit : nb * 16;
        const unsigned lane_off = (unsigned)(li * yld + 4 * kq), row_step = (unsigned)(W * yld);
        const bool lane_ok = co < q.Nc && w0 + li < W;

        // store output plane o from acc[0] (if it belongs to the segment), then rotate the accumulators
        auto retire = [&](int o) __attribute__((always_inline)) -> bool {
            const bool st = o >= dlo && o < dhi;   // wave-uniform
            if (st) {
                T* const ytile = yd + (((((int64_t)n * D + o) * H + h0) * W + w0) * yld + cbase);
                if (lane_ok) {   // the plan guarantees Nc % 4 == 0 and a pitch of whole quads: one vector store per row
#pragma unroll
                    for (int m = 0; m < MH; ++m)
                        if (h0 + m < H) {   // wave-uniform
                            const f32x4 a = acc[0][m];
                            stf4(ytile + (lane_off + (unsigned)m * row_step), make_float4(a[0] + bv.x, a[1] + bv.y, a[2] + bv.z, a[3] + bv.w));
                        }
                }
                if constexpr (STATS) {   // BatchNorm batch statistics of a = y - bias (conv_mfma.hip: same contract)
                    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
                    const bool vok = w0 + li < W;
#pragma unroll
                    for (int m = 0; m < MH; ++m) {
                        const bool ok = vok && h0 + m < H;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float a = ok ? acc[0][m][r] : 0.f;
                            s1[r] += a;
                            s2[r] += a * a;
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        s1[r] = row_sum16(s1[r]);
                        s2[r] = row_sum16(s2[r]);
                    }
                    if (li == 0) {
                        double* slot = stat_lds + ((size_t)wv * 16 + 4 * kq) * 2;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            slot[2 * r] += (double)s1[r];
                            slot[2 * r + 1] += (double)s2[r];
                        }
                    }
                }
            }
#pragma unroll
            for (int m = 0; m < MH; ++m) {
                acc[0][m] = acc[1][m];
                acc[1][m] = acc[2][m];
                acc[2][m] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            return st;
        };

        // prologue: the first PD items' pieces
        for (int k = 0; k < PD; ++k) {
            const Dma d0 = dma_open(k);
#pragma unroll
            for (int j = 0; j < MPIECES; ++j) dma_piece(d0, j);
        }
        int p = plo, c = 0;
        for (int it = 0; it < nitems; ++it) {
            // a new input plane: the previous one completed output plane p - 2 (acc[0])
            const bool stored = (c == 0 && it > 0) ? retire(p - 2) : false;
            // The DMA pieces of this item have landed once only the younger vector-memory operations are outstanding (they
            // return in order): the pieces of item it + 1 when the lead is two items, and the stores just issued.
            {
                const int younger = ((PD == 2 && it + 1 < nitems) ? 6 : 0) + (stored ? 8 : 0);
                if (younger == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                else if (younger == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                else if (younger == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
            }
            const Dma dn = dma_open(it + PD);
            const unsigned char* const buf = mybuf + (it % q.nbuf) * MBUF;
            const unsigned char* const wc = wl + c * MWCHUNK;
            const unsigned char* const wc8 = wl8 + c * MWCHUNK;

            f32x4 FA[MHH];
#pragma unroll
            for (int i = 0; i < MHH; ++i) FA[i] = *reinterpret_cast<const f32x4*>(buf + offA + i * MROW);
            f32x4 FB[MH];
            f32x2 FC[MH];
            // weight fragments: sequence 0..8 = (kd, kh) paired groups, 9..11 = kw = 2 paired group of kd, 12..14 = single tap of kd
            auto wfrag = [&](int seq) -> f32x4 {
                if (seq < 9) return *reinterpret_cast<const f32x4*>(wc + (seq / 3) * MWKD + (seq % 3) * 1024);
                if (seq < 12) return *reinterpret_cast<const f32x4*>(wc + (seq - 9) * MWKD + 3 * 1024);
                const f32x2 t = *reinterpret_cast<const f32x2*>(wc8 + (seq - 12) * MWKD);
                return f32x4{t[0], t[1], 0.f, 0.f};
            };
            f32x4 wq[3];
            wq[0] = wfrag(0);
            wq[1] = wfrag(1);
#pragma unroll
            for (int seq = 0; seq < 15; ++seq) {
                if (seq + 2 < 15) wq[(seq + 2) % 3] = wfrag(seq + 2);
                if (seq < MPIECES) dma_piece(dn, seq);
                // the kw = 2 fragments arrive while the nine (kd, kh) groups are multiplied
                if (seq >= 1 && seq <= 8) FB[seq - 1] = *reinterpret_cast<const f32x4*>(buf + offB + (seq - 1) * MROW);
                if (seq >= 9 && seq <= 12) {
                    FC[2 * (seq - 9)] = *reinterpret_cast<const f32x2*>(buf + offC + (2 * (seq - 9)) * MROW);
                    FC[2 * (seq - 9) + 1] = *reinterpret_cast<const f32x2*>(buf + offC + (2 * (seq - 9) + 1) * MROW);
                }
                __builtin_amdgcn_sched_barrier(0);   // keep the prefetches above this group's MFMAs
                const int kd = seq < 9 ? seq / 3 : (seq < 12 ? seq - 9 : seq - 12);
                const int kh = seq < 9 ? seq % 3 : 0;
                const int sl = 2 - kd;   // kd = 0 -> output plane p + 1 = acc[2]
                const f32x4 wf = wq[seq % 3];
                if (seq < 12) {
#pragma unroll
                    for (int m = 0; m < MH; ++m) {
                        const f32x4 f = seq < 9 ? FA[m + kh] : FB[m];
                        if constexpr (kBf16) {
                            acc[sl][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wf),
                                                                                 __builtin_bit_cast(bf16x8_t, f), acc[sl][m], 0, 0, 0);
                        } else {
#pragma unroll
                            for (int s4 = 0; s4 < 4; ++s4)
                                acc[sl][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[s4], f[s4], acc[sl][m], 0, 0, 0);
                        }
                    }
                } else {
#pragma unroll
                    for (int m = 0; m < MH; ++m) {
                        if constexpr (kBf16) {
                            acc[sl][m] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(
                                __builtin_bit_cast(s16x4, f32x2{wf[0], wf[1]}), __builtin_bit_cast(s16x4, FC[m]), acc[sl][m], 0, 0, 0);
                        } else {
                            acc[sl][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[0], FC[m][0], acc[sl][m], 0, 0, 0);
                            acc[sl][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[1], FC[m][1], acc[sl][m], 0, 0, 0);
                        }
                    }
                }
            }
            if (++c == q.nchunks) {
                c = 0;
                ++p;
            }
        }
        // input plane phi was the last: it completed output plane phi - 1; plane phi itself is complete when the volume ends there
        retire(phi - 1);
        retire(phi);
    }

    if constexpr (STATS) {
        __syncthreads();   // every wave of the workgroup arrives here (inactive ones went straight to it)
        for (int i = tid; i < q.Nc * 2; i += 512) {
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
    MarchGeom& q = p.q;
    q.N = g.n, q.D = g.di, q.H = g.hi, q.W = g.wi, q.Kc = Kc, q.in_ld = in_ld, q.Nc = Nc, q.out_ld = out_ld;
    const int CK = bf ? 16 : 8;
    q.nchunks = cdiv(Kc, CK);
    q.NTT = cdiv(Nc, 16);
    if (q.nchunks > 4) return false;                       // weights of one N-block + the plane buffers must fit 160 KiB of LDS
    if (stats && q.NTT > 8) return false;
    if ((int64_t)g.n * g.di * g.hi * g.wi * std::max(in_ld, out_ld) > ((int64_t)1 << 40)) return false;
    q.nbuf = q.nchunks <= 1 ? 3 : 2;
    // shape of a workgroup's eight columns: least padding first, then the squarest
    const int colsH = cdiv(g.hi, MH), colsW = cdiv(g.wi, MW);
    int best = -1, bestpad = 0;
    for (int sh = 0; sh < 4; ++sh) {
        const int gh = 1 << sh, gw = 8 >> sh;
        const int pad = cdiv(colsH, gh) * gh * cdiv(colsW, gw) * gw;
        const int sq = (sh == 1 || sh == 2) ? 1 : 0;
        if (best < 0 || pad < bestpad || (pad == bestpad && sq)) best = sh, bestpad = pad;
    }
    q.wgh = 1 << best, q.wgw = 8 >> best;
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
    const int64_t grid = wgcols * q.nseg;
    if (grid > 0x7fffffff) return false;
    p.grid = (int)grid;
    p.wp_bytes = (size_t)q.NTT * q.nchunks * MWCHUNK;
    p.smem = (size_t)((q.nchunks * MWCHUNK + 1023) & ~1023) + (size_t)MWAVES * q.nbuf * MBUF + (stats ? MWAVES * 32 * sizeof(double) : 0);
    if (p.smem > 160 * 1024) return false;
    if (force) return true;
    // the dispatcher's choice (measured on MI355X, profiles/r03_*): bf16 tensors; at least 3/4 of a round of workgroups;
    // efficiency of the (fill, halo) split at least 0.7
#if defined(MRI3D_NO_MARCH)   // tuning builds (tools/march_bench.py --lib): the round-2 dispatcher, tiled kernel everywhere
    return false;
#endif
    if (!bf) return false;
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
#define MRI3D_MARCH_CASE(STv)                                                                                           \
    if ((stat_part != nullptr) == STv) {                                                                                \
        auto kern = conv_march_kernel<T, STv>;                                                                          \
        static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                         \
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);    \
        (void)attr;                                                                                                     \
        hipLaunchKernelGGL(kern, dim3(p.grid), dim3(512), p.smem, s, (const T*)in_v, (const T*)x2,                      \
                           (const unsigned char*)ws, bias, (T*)out_v, (T*)y2, stat_part, q);                            \
    }
    MRI3D_DISPATCH_DTYPE(g.dtype, T, {
        MRI3D_MARCH_CASE(false)
        MRI3D_MARCH_CASE(true)
    });
#undef MRI3D_MARCH_CASE
    return check_launch(dgrad ? "conv3d_dgrad(march)" : "conv3d_fwd(march)");
}

}  // namespace mri3d
