// fetch_calib.hip — calibrates rocprofv3's FETCH_SIZE on the access shapes of this library (MI355X_MICROARCH.md §HBM: the
// counter is calibrated only for 16-B-per-lane coalesced streams, where it reports HALF the bytes; "other access widths are
// uncalibrated: calibrate on a known byte count in your own access pattern").  Each kernel reads every byte of a region
// exactly ONCE (regions far larger than the 256 MiB Infinity Cache, never re-read), so bytes read = a known number:
//   stream16      lane i reads 16 B at base + 16 i                                   (the guide's calibrated pattern)
//   pieces<S,P>   two lanes read one P = 32-byte piece per voxel, voxel pitch S bytes, pieces of all channel chunks
//                 visited chunk after chunk (the conv kernels' halo staging: fp32 48 ch S = 192, bf16 48 ch S = 96,
//                 bf16/fp32 16 ch S = 32 / 64)
// Build:  hipcc --offload-arch=gfx950 -O3 -o fetch_calib tools/microbench/fetch_calib.hip
// Run:    rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -o calib -- ./fetch_calib
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__global__ void stream16(const float4* __restrict__ p, float* out, size_t n) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = p[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 12345.678f) out[0] = acc;
}

// one pass over the chunk `ch` of every voxel: piece = 32 bytes = two 16-byte lane loads
template <int STRIDE>
__global__ void pieces32(const char* __restrict__ p, float* out, size_t nvox, int ch) {
    float acc = 0.f;
    const size_t nl = nvox * 2;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nl; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = *reinterpret_cast<const float4*>(p + (i >> 1) * STRIDE + ch * 32 + (i & 1) * 16);
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 12345.678f) out[0] = acc;
}

template <int STRIDE>
static void run_pieces(const char* buf, float* out, size_t bytes, const char* what) {
    const size_t nvox = bytes / STRIDE;
    for (int ch = 0; ch < STRIDE / 32; ++ch) hipLaunchKernelGGL(pieces32<STRIDE>, dim3(2048), dim3(256), 0, 0, buf, out, nvox, ch);
    hipDeviceSynchronize();
    printf("pieces32<stride %d>: %d launches (one per 32-byte chunk), %zu voxels, %.1f MB read per launch, %.1f MB in all  [%s]\n",
           STRIDE, STRIDE / 32, nvox, nvox * 32 / 1e6, nvox * (double)STRIDE / 1e6, what);
}

int main() {
    const size_t bytes = (size_t)1536 << 20;   // 1.5 GiB: six times the Infinity Cache
    char* buf;
    float* out;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&out, 256) != hipSuccess) return 1;
    hipMemset(buf, 1, bytes);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(stream16, dim3(2048), dim3(256), 0, 0, (const float4*)buf, out, bytes / 16);
    hipDeviceSynchronize();
    printf("stream16: 1 launch, %.1f MB read\n", bytes / 1e6);
    run_pieces<192>(buf, out, bytes, "fp32 48 ch, 8-channel chunks");
    run_pieces<96>(buf, out, bytes, "bf16 48 ch, 16-channel chunks");
    run_pieces<64>(buf, out, bytes, "fp32 16 ch / bf16 32 ch");
    run_pieces<32>(buf, out, bytes, "bf16 16 ch (one chunk)");
    return 0;
}
