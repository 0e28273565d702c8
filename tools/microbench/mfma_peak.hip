// mfma_peak.hip — what the fp32 MFMA pipe of this MI355X actually sustains: a register-only loop of independent
// v_mfma_f32_16x16x4_f32 (and, for comparison, v_mfma_f32_16x16x32_bf16), 2 or 4 waves per SIMD on every CU, ~1 s of work.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/mfma_peak.hip -o gpurun_out/mfma_peak && gpurun_out/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

template <int NACC>
__global__ void __launch_bounds__(256) f32_kernel(float* out, int iters, float a0, float b0) {
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = a0 + threadIdx.x, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
__global__ void __launch_bounds__(256) bf16_kernel(float* out, int iters, float a0) {
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(a0 + i); b[i] = (__bf16)(1.0f + threadIdx.x); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void __launch_bounds__(256) f32_32_kernel(float* out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    float a = a0 + threadIdx.x, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
__global__ void __launch_bounds__(256) bf16_32_kernel(float* out, int iters, float a0) {
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    bf16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(a0 + i); b[i] = (__bf16)(1.0f + threadIdx.x); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// multi-block shapes: v_mfma_f32_16x16x1_4b_f32 (four independent 16x16 rank-1 updates, 16 accumulator registers)
template <int NACC>
__global__ void __launch_bounds__(256) f32_16x1_kernel(float* out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    float a = a0 + threadIdx.x, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x1f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
static double time_ms(F launch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    float* out;
    hipMalloc(&out, 4096 * 256 * sizeof(float));
    const int iters = 200000;   // ~50-100 ms per launch: long enough for the power management to settle
    for (int wgs_per_cu = 2; wgs_per_cu <= 4; wgs_per_cu += 2) {
        const int grid = 256 * wgs_per_cu;
        {
            double ms = time_ms([&] { hipLaunchKernelGGL(f32_kernel<8>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f, 2.0f); });
            double flops = (double)grid * 4 * iters * 8 * 2048.0;
            printf("fp32 16x16x4  %d waves/SIMD: %.2f ms  %.1f TFLOP/s\n", wgs_per_cu, ms, flops / ms / 1e9);
        }
        {
            double ms = time_ms([&] { hipLaunchKernelGGL(bf16_kernel<8>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f); });
            double flops = (double)grid * 4 * iters * 8 * 16384.0;
            printf("bf16 16x16x32 %d waves/SIMD: %.2f ms  %.1f TFLOP/s\n", wgs_per_cu, ms, flops / ms / 1e9);
        }
        {
            double ms = time_ms([&] { hipLaunchKernelGGL(f32_32_kernel<4>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f, 2.0f); });
            double flops = (double)grid * 4 * iters * 4 * 4096.0;
            printf("fp32 32x32x2  %d waves/SIMD: %.2f ms  %.1f TFLOP/s\n", wgs_per_cu, ms, flops / ms / 1e9);
        }
        {
            double ms = time_ms([&] { hipLaunchKernelGGL(f32_16x1_kernel<4>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f, 2.0f); });
            double flops = (double)grid * 4 * iters * 4 * 2048.0;
            printf("fp32 16x16x1_4b %d waves/SIMD: %.2f ms  %.1f TFLOP/s\n", wgs_per_cu, ms, flops / ms / 1e9);
        }
        {   // dependency distance: the same instruction with only 2 / 1 independent accumulators
            double ms = time_ms([&] { hipLaunchKernelGGL(f32_16x1_kernel<2>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f, 2.0f); });
            printf("fp32 16x16x1_4b %d waves/SIMD, 2 accumulators: %.1f TFLOP/s\n", wgs_per_cu, (double)grid * 4 * iters * 2 * 2048.0 / ms / 1e9);
            ms = time_ms([&] { hipLaunchKernelGGL(f32_16x1_kernel<1>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f, 2.0f); });
            printf("fp32 16x16x1_4b %d waves/SIMD, 1 accumulator:  %.1f TFLOP/s\n", wgs_per_cu, (double)grid * 4 * iters * 1 * 2048.0 / ms / 1e9);
#define F32_SWEEP(NA)                                                                                                  \
    ms = time_ms([&] { hipLaunchKernelGGL(f32_kernel<NA>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f, 2.0f); });              \
    printf("fp32 16x16x4    %d waves/SIMD, %d accumulators: %.1f TFLOP/s\n", wgs_per_cu, NA, (double)grid * 4 * iters * NA * 2048.0 / ms / 1e9);
            F32_SWEEP(1) F32_SWEEP(2) F32_SWEEP(3) F32_SWEEP(4) F32_SWEEP(5) F32_SWEEP(6) F32_SWEEP(7) F32_SWEEP(16) F32_SWEEP(28)
        }
        {
            double ms = time_ms([&] { hipLaunchKernelGGL(bf16_32_kernel<4>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f); });
            double flops = (double)grid * 4 * iters * 4 * 32768.0;
            printf("bf16 32x32x16 %d waves/SIMD: %.2f ms  %.1f TFLOP/s\n", wgs_per_cu, ms, flops / ms / 1e9);
        }
    }
    hipFree(out);
    return 0;
}
