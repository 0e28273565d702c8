// mfma_valu_mix.hip — does other work hide behind fp32 MFMAs?  Per loop iteration: 16 independent v_mfma_f32_16x16x4_f32
// plus V vector FMAs (or V LDS reads), all independent.  If the extra instructions are hidden the time does not change
// with V; if they are additive it grows by ~V * 4 clk per iteration.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/mfma_valu_mix.hip -o gpurun_out/mfma_valu_mix && gpurun_out/mfma_valu_mix
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int V, bool LDS>
__global__ void __launch_bounds__(256) mix_kernel(float* out, int iters, float a0, float b0) {
    __shared__ float sh[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) sh[i] = (float)i;
    __syncthreads();
    f32x4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float v[V > 0 ? V : 1];
#pragma unroll
    for (int i = 0; i < (V > 0 ? V : 1); ++i) v[i] = a0 + i;
    float a = a0 + threadIdx.x, b = b0;
    int idx = threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
            // V/16 extra instructions after every MFMA
#pragma unroll
            for (int j = i * V / 16; j < (i + 1) * V / 16; ++j) {
                if (LDS) v[j] += sh[(idx + 64 * j) & 4095];
                else v[j] = fmaf(v[j], 1.0001f, 0.5f);
            }
        }
        idx += 7;
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
#pragma unroll
    for (int i = 0; i < (V > 0 ? V : 1); ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// S scalar-ALU instructions (s_mul_i32 / s_add_i32 on four independent scalar registers) per 16 MFMAs: the staging address
// arithmetic of the conv kernels is mostly scalar (239 SALU instructions per 448 MFMAs in conv_mfma_fwd2_kernel<float,1>)
template <int S>
__global__ void __launch_bounds__(256) mix_salu_kernel(float* out, int iters, float a0, float b0) {
    f32x4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = a0 + threadIdx.x, b = b0;
    int s0 = iters, s1 = iters + 1, s2 = iters + 2, s3 = iters + 3;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
#pragma unroll
            for (int j = i * S / 16; j < (i + 1) * S / 16; ++j) {
                // one SALU instruction each; the empty asm keeps the compiler from folding the chain
                if ((j & 3) == 0) { s0 = s0 * 3; asm volatile("" : "+s"(s0)); }
                else if ((j & 3) == 1) { s1 = s1 + 7; asm volatile("" : "+s"(s1)); }
                else if ((j & 3) == 2) { s2 = s2 * 5; asm volatile("" : "+s"(s2)); }
                else { s3 = s3 + 9; asm volatile("" : "+s"(s3)); }
            }
        }
    }
    float s = (float)(s0 + s1 + s2 + s3);
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename K>
static double run(K kern, float* out, int grid, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f, 2.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f, 2.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) printf("  launch error: %s\n", hipGetErrorString(err));
    return ms;
}

int main() {
    float* out;
    hipMalloc(&out, 2048 * 256 * sizeof(float));
    const int iters = 50000, grid = 512;   // 2 waves per SIMD
    const double mfma_flops = (double)grid * 4 * iters * 16 * 2048.0;
#define ROW(V, L)                                                                                                   \
    {                                                                                                               \
        double ms = run(mix_kernel<V, L>, out, grid, iters);                                                        \
        printf("%s x %2d per 16 MFMAs: %.2f ms  MFMA rate %.1f TFLOP/s\n", L ? "ds_read_b32" : "v_fma_f32  ", V, ms, mfma_flops / ms / 1e9); \
    }
    ROW(0, false) ROW(8, false) ROW(16, false) ROW(32, false) ROW(64, false) ROW(128, false)
    ROW(8, true) ROW(16, true) ROW(32, true) ROW(64, true)
#define SROW(S)                                                                                                     \
    {                                                                                                               \
        double ms = run(mix_salu_kernel<S>, out, grid, iters);                                                      \
        printf("s_mul/s_add  x %3d per 16 MFMAs: %.2f ms  MFMA rate %.1f TFLOP/s\n", S, ms, mfma_flops / ms / 1e9);       \
    }
    SROW(0) SROW(8) SROW(16) SROW(32) SROW(64) SROW(128)
    hipFree(out);
    return 0;
}
