// mfma_mem_mix.hip — what does a register-filling memory instruction cost an fp32-MFMA-bound wave?
// Per loop iteration: 16 independent v_mfma_f32_16x16x4_f32 whose A operands are the registers filled by the PREVIOUS
// iteration's R reads (double-buffered, as in the conv kernels), plus R reads of one kind for the next iteration:
//   ds_read_b128 / ds_read_b64 / ds_read_b32 (conflict-free, lane-linear) or global_load_dwordx4 (a 64 KB buffer: L1/L2 hits).
// 2 waves per SIMD (512 workgroups of 256 threads).  Time per iteration minus the R = 0 time, divided by R = the cost of one read
// in MFMA-pipe cycles.  Build: hipcc --offload-arch=gfx950 -O3 -o mfma_mem_mix tools/microbench/mfma_mem_mix.hip
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// KIND: 0 = ds_read_b128, 1 = ds_read_b64, 2 = ds_read_b32, 3 = global_load_dwordx4
template <int R, int KIND>
__global__ void __launch_bounds__(256, 2) mix_kernel(float* out, const float4* __restrict__ gsrc, int iters, float b0) {
    __shared__ __attribute__((aligned(16))) float sh[16384];   // 64 KB
    for (int i = threadIdx.x; i < 16384; i += 256) sh[i] = (float)(i & 255) * 0.001f;
    __syncthreads();
    f32x4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int RR = R > 0 ? R : 1;
    f32x4 cur[RR], nxt[RR];
#pragma unroll
    for (int r = 0; r < RR; ++r) cur[r] = f32x4{1.f + threadIdx.x, 2.f, 3.f, 4.f};
    const float b = b0;
    int off = threadIdx.x;   // lane-linear: conflict-free for every width
    for (int it = 0; it < iters; ++it) {
        if (R > 0) {
#pragma unroll
            for (int r = 0; r < RR; ++r) {
                const int o = (off + 256 * r) & 4095;   // 16-byte units inside the 64 KB array / buffer
                if (KIND == 0) nxt[r] = *reinterpret_cast<const f32x4*>(sh + 4 * o);
                else if (KIND == 1) { const float2 t = *reinterpret_cast<const float2*>(sh + 4 * o); nxt[r] = f32x4{t.x, t.y, t.x, t.y}; }
                else if (KIND == 2) { const float t = sh[4 * o]; nxt[r] = f32x4{t, t, t, t}; }
                else { const float4 t = gsrc[o]; nxt[r] = f32x4{t.x, t.y, t.z, t.w}; }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[i % RR][i & 3], b, acc[i], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (R > 0) {
#pragma unroll
            for (int r = 0; r < RR; ++r) cur[r] = nxt[r];
        }
        off += 64;
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename K>
static double run(K kern, float* out, const float4* g, int grid, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, g, iters, 2.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, g, iters, 2.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    float* out;
    float4* g;
    hipMalloc(&out, 512 * 256 * sizeof(float));
    hipMalloc(&g, 65536);
    hipMemset(g, 0, 65536);
    const int iters = 20000, grid = 512;
    const double flops = (double)grid * 4 * iters * 16 * 2048.0;
    double base = 0;
    const char* names[4] = {"ds_read_b128", "ds_read_b64 ", "ds_read_b32 ", "global_load_dwordx4"};
#define ROW(R, K)                                                                                                     \
    {                                                                                                                 \
        const double ms = run(mix_kernel<R, K>, out, g, grid, iters);                                                 \
        if (R == 0) base = ms;                                                                                        \
        /* one SIMD runs 2 waves: iterations per SIMD = 2 * iters; MFMA-pipe clocks per iteration at 2.4 GHz */        \
        const double clk_per_it = ms * 1e-3 * 2.4e9 / (2.0 * iters);                                                  \
        printf("%-20s x %d per 16 MFMAs: %7.2f ms  %6.1f TFLOP/s  %6.0f clk/iteration  (+%.1f clk per read)\n", R ? names[K] : "no reads", R, ms, \
               flops / ms / 1e9, clk_per_it, R ? (ms - base) * 1e-3 * 2.4e9 / (2.0 * iters) / R : 0.0);                \
    }
    ROW(0, 0)
    ROW(1, 0) ROW(2, 0) ROW(4, 0) ROW(8, 0)
    ROW(2, 1) ROW(4, 1) ROW(8, 1)
    ROW(2, 2) ROW(4, 2) ROW(8, 2)
    ROW(1, 3) ROW(2, 3) ROW(4, 3) ROW(8, 3)
    return 0;
}
