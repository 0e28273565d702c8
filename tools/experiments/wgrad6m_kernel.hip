// NOT part of the library: the fp32 counterpart of conv_mfma_wgrad_bf16m_kernel (weight gradient marching along d), as it was
// built, tested (fuzz + operator suites green) and measured in round 2 inside csrc/conv_mfma.hip (it uses that file's
// constants MTH, MXP, MXS, MYP, MYS, MHP, FTW, DLS, rot_slot, ldg4, g_zero16, TileWalk).  Result against the tile kernel (v6), same box:
//   16 -> 16 @160x192x160 x2   1.204 -> 1.249 ms        48 -> 16 (cat)   3.354 -> 3.399 ms       96 -> 32 (cat) @80x96x80  1.834 -> 1.806 ms
//   32 -> 32 @80x96x80 x2      0.619 -> 0.673 ms        16 -> 16 @32^3 x512   2.206 -> 2.030 ms
// The fp32 weight gradient is MFMA-bound (81 % busy) at 256 registers: staging fewer bytes does not pay there.  Kept for the record.
// ------------------------------------------------------------------ fp32 weight gradient, marching along d
// conv_mfma_wgrad_bf16m_kernel's structure for fp32 tensors (16-channel operands): a column of 8 rows x 16 voxels, a four-plane
// LDS ring of X, dY double-buffered, loads two planes ahead.  The tile kernel (v6) is MFMA-bound, but its staging — 13 loads,
// the register transposes and 13 LDS writes per lane and 2 x 6-row tile, for 32 rows of X per 12 output rows — costs it 20 %
// (no-load ablation, 16 -> 16: 0.675 -> 0.540 ms); marching stages 10 rows of X per 8 output rows and needs no border path:
// every per-lane offset and validity is fixed for the whole column.  Fragments, K permutation, accumulators and partial
// layout are v6's.  Staging units are 4 voxels x 4 channels (v6's): 160 of X and 128 of dY per plane plus 64 halo voxels —
// every lane owns unit `tid`, lanes 0..31 a second one, lanes 64..127 a halo voxel.
template <bool BIAS>
__global__ void __launch_bounds__(256, 2)
conv_mfma_wgrad6m_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ part, int N, int D,
                         int H, int W, int Ci, int x_ld, int Co, int y_ld, int nseg, int tilesH, int tilesW, int ntasks,
                         const float* __restrict__ x2, int x2_ld, int ksplit, int segl) {
    constexpr int TG = 27, TGA = TG + (BIAS ? 1 : 0);
    int xc0 = (int)blockIdx.y * 16;
    if (x2 != nullptr && xc0 >= ksplit) { x = x2; x_ld = x2_ld; xc0 -= ksplit; }   // conv over cat((x, x2)): the ci-tile's tensor
    extern __shared__ __attribute__((aligned(16))) float lds[];
    char* xs = reinterpret_cast<char*>(lds);
    char* ys = xs + MXS;
    char* yh = ys + MYS;
    const int cit = blockIdx.y, cob = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kq = lane >> 4;

    f32x4 acc[TGA];
#pragma unroll
    for (int t = 0; t < TGA; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

    // staging slots of the lane.  Slot 0: unit `tid` (X units 0..159, then dY units 0..95).  Slot 1: dY units 96..127 on lanes
    // 0..31, the 64 halo voxels on lanes 64..127, nothing elsewhere.  kind: 0 X unit, 1 dY unit, 2 halo voxel, 3 none.
    const int kind0 = tid < 160 ? 0 : 1;
    const int u0 = tid < 160 ? tid : tid - 160;
    const int kind1 = tid < 32 ? 1 : (tid >= 64 && tid < 128) ? 2 : 3;
    const int u1 = tid < 32 ? 96 + tid : tid - 64;
    // operand addresses of the wave's two output rows (v6's maps)
    const int orow0 = wv * 2;
    const int xline0 = (orow0 * 16 + li) * DLS + 16 * rot_slot(kq, li);                 // + slot * MXP + kh * 16 * DLS
    const int yline0 = orow0 * 16 + li;
    const int yrow_o = yline0 * DLS + 16 * rot_slot(kq, li);                            // + buffer * MYP
    const bool pl_halo = kq == 0, nr_halo = kq == 3;
    const int ypl_o = pl_halo ? yline0 * 8 : yline0 * DLS + 16 * rot_slot(kq - 1, li) + 12;
    const int ynr_o = nr_halo ? yline0 * 8 + 4 : yline0 * DLS + 16 * rot_slot(kq + 1, li);
    const int pl_step = pl_halo ? 16 * 8 : 16 * DLS, nr_step = nr_halo ? 16 * 8 : 16 * DLS;

    float4 v[2][2][4];   // [register set][slot][voxel]
    const TileWalk tw = tile_walk(ntasks);
    for (int k = 0; k < tw.count; ++k) {
        int task = tw.first + k * tw.stride;
        const int w0 = (task % tilesW) * FTW;
        task /= tilesW;
        const int h0 = (task % tilesH) * MTH;
        task /= tilesH;
        const int seg = task % nseg, n = task / nseg;
        const int dA = seg * segl, dB = min(D, dA + segl);
        // per slot: element offset inside a plane and the number of valid voxels (0..4; halo 0 / 1), fixed for the column
        int off[2] = {0, 0}, nval[2] = {0, 0};
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
            const int kind = sl == 0 ? kind0 : kind1, uu = sl == 0 ? u0 : u1;
            if (kind == 0) {
                const int q = uu & 3, wg = (uu >> 2) & 3, row = uu >> 4;
                const int gh = h0 - 1 + row;
                if ((unsigned)gh < (unsigned)H) {   // Ci % 16 == 0 (host): every channel quad exists
                    off[sl] = (gh * W + w0 + 4 * wg) * x_ld + xc0 + 4 * q;
                    nval[sl] = min(4, max(0, W - (w0 + 4 * wg)));
                }
            } else if (kind == 1) {
                const int q = uu & 3, wg = (uu >> 2) & 3, row = uu >> 4;
                const int gh = h0 + row, c0 = cob * 16 + 4 * q;
                if (gh < H && c0 < Co) {            // Co % 4 == 0 (host)
                    off[sl] = (gh * W + w0 + 4 * wg) * y_ld + c0;
                    nval[sl] = min(4, max(0, W - (w0 + 4 * wg)));
                }
            } else if (kind == 2) {
                const int q = uu & 3, side = (uu >> 2) & 1, row = uu >> 3;
                const int gh = h0 + row, gw = side ? w0 + FTW : w0 - 1, c0 = cob * 16 + 4 * q;
                if (gh < H && (unsigned)gw < (unsigned)W && c0 < Co) {
                    off[sl] = (gh * W + gw) * y_ld + c0;
                    nval[sl] = 1;
                }
            }
        }
        const int64_t xplane = (int64_t)H * W * x_ld, yplane = (int64_t)H * W * y_ld;
        const float* const xcol = x + (int64_t)n * D * xplane;
        const float* const ycol = dy + (int64_t)n * D * yplane;

        auto step = [&](auto SETC, int t) {
            constexpr int S = decltype(SETC)::value;
            // ---- write what step t-2 loaded: X plane t+1, dY plane t (zeros where a plane or voxel does not exist)
            if (t >= dA - 2 && t <= dB) {
#pragma unroll
                for (int sl = 0; sl < 2; ++sl) {
                    const int kind = sl == 0 ? kind0 : kind1, uu = sl == 0 ? u0 : u1;
                    const float4* vv = v[S][sl];
                    if (kind == 0 || kind == 1) {
                        const int q = uu & 3, wg = (uu >> 2) & 3, row = uu >> 4;
                        char* dst = (kind == 0 ? xs + ((t + 1) & 3) * MXP : ys + (t & 1) * MYP) + (row * 16 + 4 * q) * DLS +
                                    16 * rot_slot(wg, 4 * q);   // 4 channels share a rotation
                        *reinterpret_cast<float4*>(dst) = make_float4(vv[0].x, vv[1].x, vv[2].x, vv[3].x);
                        *reinterpret_cast<float4*>(dst + DLS) = make_float4(vv[0].y, vv[1].y, vv[2].y, vv[3].y);
                        *reinterpret_cast<float4*>(dst + 2 * DLS) = make_float4(vv[0].z, vv[1].z, vv[2].z, vv[3].z);
                        *reinterpret_cast<float4*>(dst + 3 * DLS) = make_float4(vv[0].w, vv[1].w, vv[2].w, vv[3].w);
                    } else if (kind == 2) {
                        const int q = uu & 3, side = (uu >> 2) & 1, row = uu >> 3;
                        char* dst = yh + (t & 1) * MHP + (row * 16 + 4 * q) * 8 + 4 * side;
                        *reinterpret_cast<float*>(dst) = vv[0].x;
                        *reinterpret_cast<float*>(dst + 8) = vv[0].y;
                        *reinterpret_cast<float*>(dst + 16) = vv[0].z;
                        *reinterpret_cast<float*>(dst + 24) = vv[0].w;
                    }
                }
            }
            // ---- load X plane t+3 / dY plane t+2 (address select: a missing piece reads the zero block; nothing waits here)
#pragma unroll
            for (int sl = 0; sl < 2; ++sl) {
                const int kind = sl == 0 ? kind0 : kind1;
                const int pl = kind == 0 ? t + 3 : t + 2;
                const bool pok = kind == 0 ? (pl >= dA - 1 && pl <= dB && (unsigned)pl < (unsigned)D) : (pl >= dA && pl < dB);
                const float* src = (kind == 0 ? xcol + (int64_t)pl * xplane : ycol + (int64_t)pl * yplane) + off[sl];
                const int ldv = kind == 0 ? x_ld : y_ld;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[S][sl][j] = ldg4((pok && j < nval[sl]) ? src + (int64_t)j * ldv : g_zero16);
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---- multiply plane p = t - 1
            const int p = t - 1;
            if (p >= dA && p < dB) {
                const char* const xb0 = xs + ((p - 1) & 3) * MXP + xline0;
                const char* const xb1 = xs + (p & 3) * MXP + xline0;
                const char* const xb2 = xs + ((p + 1) & 3) * MXP + xline0;
                const char* yrow = ys + (p & 1) * MYP + yrow_o;
                const char* ypl = (pl_halo ? yh + (p & 1) * MHP : ys + (p & 1) * MYP) + ypl_o;
                const char* ynr = (nr_halo ? yh + (p & 1) * MHP : ys + (p & 1) * MYP) + ynr_o;
#pragma unroll 1
                for (int r = 0; r < 2; ++r, yrow += 16 * DLS, ypl += pl_step, ynr += nr_step) {
                    const float4 b = *reinterpret_cast<const float4*>(yrow);
                    const float plv = *reinterpret_cast<const float*>(ypl);   // dY[first - 1]
                    const float nrv = *reinterpret_cast<const float*>(ynr);   // dY[last + 1]
                    const float b0[4] = {b.y, b.z, b.w, nrv}, b1[4] = {b.x, b.y, b.z, b.w}, b2[4] = {plv, b.x, b.y, b.z};   // dY[u+1], dY[u], dY[u-1]
                    if constexpr (BIAS) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[TG] = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, b1[j], acc[TG], 0, 0, 0);
                    }
                    const int rowo = r * 16 * DLS;
#pragma unroll
                    for (int kdh = 0; kdh < 9; ++kdh) {
                        const char* xb = kdh < 3 ? xb0 : (kdh < 6 ? xb1 : xb2);
                        const float4 g = *reinterpret_cast<const float4*>(xb + rowo + (kdh % 3) * 16 * DLS);
                        const float a[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            acc[kdh * 3 + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b0[j], acc[kdh * 3 + 0], 0, 0, 0);
                            acc[kdh * 3 + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b1[j], acc[kdh * 3 + 1], 0, 0, 0);
                            acc[kdh * 3 + 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b2[j], acc[kdh * 3 + 2], 0, 0, 0);
                        }
                    }
                }
            }
            __syncthreads();   // this step's LDS writes are visible, its reads are done
        };
        const int t0 = (dA - 4) & ~1;   // steps dA-4 .. dB, two per iteration: the register set (t & 1) is a compile-time index
        for (int t = t0; t <= dB; t += 2) {
            step(std::integral_constant<int, 0>{}, t);
            step(std::integral_constant<int, 1>{}, t + 1);
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

