// common.h — shared host/device helpers for libmri3d_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <stdlib.h>
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

// A/B switches exist only in a tuning build (-DMRI3D_TUNING, tools/): the shipped library reads no environment variable, so a
// stray variable in a job's environment cannot change which kernel runs or what it computes.
#ifdef MRI3D_TUNING
inline int tuning_knob(const char* name, int dflt) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}
#else
constexpr int tuning_knob(const char*, int dflt) { return dflt; }
#endif

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

// ---------------------------------------------------------------- storage types
// Activations are stored as float (MRI3D_F32) or bfloat16 (MRI3D_BF16, BASELINE configs[3]); arithmetic, statistics,
// parameters and parameter gradients are always fp32.  gfx950 converts with v_cvt_pk_bf16_f32 (round-to-nearest-even).
typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;

template <typename T> __device__ __forceinline__ float ldf(const T* p) { return (float)*p; }
template <typename T> __device__ __forceinline__ void stf(T* p, float v) { *p = (T)v; }

// 4 consecutive elements; p must be aligned to 4 elements (16 B for float, 8 B for bf16)
__device__ __forceinline__ float4 ldf4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 ldf4(const bf16_t* p) {
    const bf16x4_t v = *reinterpret_cast<const bf16x4_t*>(p);
    return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
}
__device__ __forceinline__ void stf4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void stf4(bf16_t* p, float4 v) {
    bf16x4_t o;
    o[0] = (bf16_t)v.x, o[1] = (bf16_t)v.y, o[2] = (bf16_t)v.z, o[3] = (bf16_t)v.w;
    *reinterpret_cast<bf16x4_t*>(p) = o;
}

inline size_t dtype_size(int dtype) { return dtype == MRI3D_BF16 ? 2 : 4; }
// all pointers aligned for 4-element vector access in the given storage type
inline bool aligned_vec4(int dtype, const void* a, const void* b = nullptr, const void* c = nullptr) {
    const uintptr_t m = dtype == MRI3D_BF16 ? 7 : 15;
    return ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c)) & m) == 0;
}

// every given pointer 16-byte aligned (dwordx4 accesses, LDS-DMA pieces)
inline bool aligned16(const void* a, const void* b = nullptr, const void* c = nullptr) {
    return ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c)) & 15) == 0;
}

// Run `body` with `T` bound to the storage type of `dtype`.
#define MRI3D_DISPATCH_DTYPE(dtype, T, ...)  \
    do {                                     \
        if ((dtype) == MRI3D_BF16) {         \
            using T = ::mri3d::bf16_t;       \
            __VA_ARGS__                      \
        } else {                             \
            using T = float;                 \
            __VA_ARGS__                      \
        }                                    \
    } while (0)

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
