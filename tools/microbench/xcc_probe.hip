// Which XCD does workgroup (x, y, z) of a 3-D grid land on?  (tile_walk in conv_mfma.hip assumes linear id % 8.)
//   hipcc --offload-arch=gfx950 -O2 tools/microbench/xcc_probe.hip -o tools/microbench/xcc_probe && tools/microbench/xcc_probe 170 3 1
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void probe(int* out) {
    if (threadIdx.x == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        const int id = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        out[2 * id] = (int)(xcc & 0xf);
        out[2 * id + 1] = (int)hwid;
    }
    // keep the workgroup resident for a while so that the whole grid is in flight together
    for (int i = 0; i < 200; ++i) __builtin_amdgcn_s_sleep(100);
}
int main(int argc, char** argv) {
    const int gx = argc > 1 ? atoi(argv[1]) : 170, gy = argc > 2 ? atoi(argv[2]) : 3, gz = argc > 3 ? atoi(argv[3]) : 1;
    const int n = gx * gy * gz;
    int* d;
    hipMalloc(&d, 2 * n * sizeof(int));
    hipLaunchKernelGGL(probe, dim3(gx, gy, gz), dim3(256), 65536, 0, d);
    std::vector<int> h(2 * n);
    hipMemcpy(h.data(), d, 2 * n * sizeof(int), hipMemcpyDeviceToHost);
    int match = 0;
    for (int i = 0; i < n; ++i) match += h[2 * i] == i % 8;
    printf("grid %d x %d x %d: %d of %d workgroups on XCD (linear id %% 8)\n", gx, gy, gz, match, n);
    for (int y = 0; y < gy * gz; ++y) {
        printf("row %d: ", y);
        for (int x = 0; x < 24 && x < gx; ++x) printf("%d ", h[2 * (x + gx * y)]);
        printf("\n");
    }
    return 0;
}
