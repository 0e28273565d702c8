// common.h — shared host/device helpers for libmri3d_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/mri3d.h"

namespace mri3d {

void set_error(const char* fmt, ...);

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return MRI3D_ELAUNCH;
    }
    return MRI3D_OK;
}

#define MRI3D_REQUIRE(cond, code, ...)      \
    do {                                    \
        if (!(cond)) {                      \
            ::mri3d::set_error(__VA_ARGS__); \
            return (code);                  \
        }                                   \
    } while (0)

inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline int cdiv(int a, int b) { return (a + b - 1) / b; }
inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Cap for grid-stride streaming kernels: 256 CUs x 8 blocks (cdna_hip_programming.md Guideline 11).
constexpr int kMaxStreamBlocks = 2048;

inline int stream_grid(int64_t work_items, int per_block) {
    int64_t b = cdiv64(work_items, per_block);
    if (b > kMaxStreamBlocks) b = kMaxStreamBlocks;
    if (b < 1) b = 1;
    return (int)b;
}

// ---------------------------------------------------------------- device helpers
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__device__ __forceinline__ float apply_act(float v, int act, float slope) {
    // act: MRI3D_ACT_*; for PReLU the caller passes alpha as `slope`.
    if (act == MRI3D_ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == MRI3D_ACT_LEAKY || act == MRI3D_ACT_PRELU) return v > 0.f ? v : v * slope;
    return v;
}

}  // namespace mri3d
