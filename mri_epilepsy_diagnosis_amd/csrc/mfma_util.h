// mfma_util.h — device helpers shared by the MFMA convolution kernels (conv_mfma.hip, conv_march.hip); gfx950 only.
#pragma once
#include "common.h"

namespace mri3d {

typedef float f32x4 __attribute__((ext_vector_type(4)));

typedef int i32x4 __attribute__((ext_vector_type(4)));

// One LDS-DMA piece (`buffer_load_dwordx4 ... offen lds`): lane l's 16 bytes at byte offset `voff` of the buffer resource `rs`
// land at LDS byte address lds_dst + 16*l (lds_dst wave-uniform: a wave-instruction fills 1 KiB); an offset >= num_records
// writes ZEROS.  No VGPR destination, no ds_write.  Semantics probed on the hardware by tools/microbench/lds_dma_probe.hip
// (incl. LDS addresses above 64 KiB).  The load is invisible to hipcc's s_waitcnt bookkeeping: the kernel waits for its DMA
// pieces itself (a counted vmcnt in front of the chunk barrier); hipcc's own counts stay conservative-correct because VMEM
// returns in order.  M0 (the LDS base of the DMA) is compiler-reserved: saved and restored inside the statement.
__device__ __forceinline__ void lds_dma16(unsigned voff, i32x4 rs, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(rs), "s"(lds_dst)
                 : "memory");
}
constexpr unsigned kDmaOob = 0xffffff80u;      // >= kDmaRecords: an out-of-volume piece (zeros)
constexpr unsigned kDmaRecords = 0xffffff00u;  // num_records of the per-item resource (its base is the item's halo origin)


// Bijective XCD-aware remap (cdna_hip_programming.md T1): blocks b and b+8 share an XCD; give each XCD a contiguous
// range of tiles so that spatial neighbours (which share halo voxels) hit the same L2.
__device__ __forceinline__ int xcd_remap(int b, int n) {
    const int q = n >> 3, r = n & 7, xcd = b & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
}


// sum over the 16 lanes of a DPP row (lanes with equal lane >> 4): four row rotations, every lane ends with the total
__device__ __forceinline__ float row_sum16(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));   // row_ror:8
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));   // row_ror:4
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));   // row_ror:2
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));   // row_ror:1
    return v;
}


}  // namespace mri3d
