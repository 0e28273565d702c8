// lds_dma_probe.hip — semantics of `buffer_load_dwordx4 … offen lds` on gfx950 that the conv staging relies on (round 2):
//   (1) lane l of a wave-instruction writes 16 bytes at M0 + 16*l (a wave writes 1 KiB contiguously);
//   (2) a lane whose offset is >= num_records of the buffer resource writes ZEROS (out-of-volume halo pieces);
//   (3) M0 may point above 64 KiB (the second halo buffer of conv_mfma_fwd2_kernel starts at 36 KiB and ends at 72 KiB;
//       probed up to 150 KiB).
// hipcc --offload-arch=gfx950 -O3 -o lds_dma_probe tools/microbench/lds_dma_probe.hip && ./lds_dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int i32x4 __attribute__((ext_vector_type(4)));

__global__ void probe(const float* x, float* y, int nbytes, int lds_base_bytes) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, wv = tid >> 6;
    for (int i = tid; i < 256 * 4; i += 256) lds[lds_base_bytes / 4 + i] = -1.f;   // poison
    __syncthreads();
    i32x4 rs;
    const unsigned long long p = (unsigned long long)x;
    rs[0] = (int)(p & 0xffffffffu);
    rs[1] = (int)(p >> 32);
    rs[2] = nbytes;
    rs[3] = 0x00020000;
    // lane l reads piece (255 - tid) (a per-lane SOURCE permutation); every 5th lane is "out of volume"
    unsigned voff = (unsigned)(255 - tid) * 16u;
    if (tid % 5 == 0) voff = 0xfffffff0u;
    const unsigned dst = (unsigned)(size_t)lds + lds_base_bytes + wv * 1024;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rs), "s"(__builtin_amdgcn_readfirstlane(dst)) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = tid; i < 256 * 4; i += 256) y[i] = lds[lds_base_bytes / 4 + i];
}

int main() {
    const int n = 256 * 4;
    std::vector<float> hx(n), hy(n);
    for (int i = 0; i < n; ++i) hx[i] = (float)(i + 1);
    float *dx, *dy;
    hipMalloc(&dx, n * 4);
    hipMalloc(&dy, n * 4);
    hipMemcpy(dx, hx.data(), n * 4, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    int bad_total = 0;
    for (int base : {0, 36864, 65536 - 2048, 65536, 100 * 1024, 150 * 1024}) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(256), base + 4096, 0, dx, dy, n * 4, base);
        if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed at base %d\n", base); return 1; }
        hipMemcpy(hy.data(), dy, n * 4, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int t = 0; t < 256; ++t)
            for (int s = 0; s < 4; ++s) {
                const float want = (t % 5 == 0) ? 0.f : hx[(255 - t) * 4 + s];
                if (hy[t * 4 + s] != want) {
                    if (bad < 4) printf("  base %d lane %d s %d: got %g want %g\n", base, t, s, hy[t * 4 + s], want);
                    ++bad;
                }
            }
        printf("LDS base %6d B: %s (%d mismatches)\n", base, bad ? "FAIL" : "ok: lane-linear 16-byte pieces, out-of-range lanes write zeros", bad);
        bad_total += bad;
    }
    return bad_total ? 1 : 0;
}
