// elementwise.hip — channel-slice copy / add (torch.cat and residual sums on NDHWC buffers with a voxel pitch) and
// the flat-buffer Adam/AdamW step.  Reference: torch.cat in unet.UNet's decoder and modified_3dunet.py:158-178;
// `out += residual` modified_3dunet.py:108; torch.optim.AdamW segmentation/routine.py:358; Adam classification/routine.py:271.
// All HBM-bound streaming kernels (16 B per lane when alignment allows).
#include "common.h"

namespace mri3d {

template <int VEC>
__global__ void __launch_bounds__(256)
copy_channels_kernel(const float* __restrict__ src, float* __restrict__ dst, int64_t nvox, int C, int s_ld, int d_ld) {
    const int CV = C / VEC;
    const int64_t total = nvox * CV;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int cv = (int)(i % CV);
        int64_t v = i / CV;
        if (VEC == 4)
            *reinterpret_cast<float4*>(dst + v * d_ld + cv * 4) = *reinterpret_cast<const float4*>(src + v * s_ld + cv * 4);
        else
            dst[v * d_ld + cv] = src[v * s_ld + cv];
    }
}

template <int VEC>
__global__ void __launch_bounds__(256)
add_channels_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ dst, int64_t nvox,
                    int C, int a_ld, int b_ld, int d_ld) {
    const int CV = C / VEC;
    const int64_t total = nvox * CV;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int cv = (int)(i % CV);
        int64_t v = i / CV;
        if (VEC == 4) {
            float4 x = *reinterpret_cast<const float4*>(a + v * a_ld + cv * 4);
            float4 y = *reinterpret_cast<const float4*>(b + v * b_ld + cv * 4);
            *reinterpret_cast<float4*>(dst + v * d_ld + cv * 4) = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
        } else {
            dst[v * d_ld + cv] = a[v * a_ld + cv] + b[v * b_ld + cv];
        }
    }
}

// torch.optim.Adam(W) single-tensor update, bias-corrected, eps added after the sqrt(v_hat)
__global__ void __launch_bounds__(256)
adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, int64_t n,
            float lr, float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt, float gscale, int decoupled) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float pi = p[i];
        float gi = g[i] * gscale;
        if (decoupled) pi *= (1.f - lr * wd);
        else gi = fmaf(wd, pi, gi);
        float mi = fmaf(b1, m[i], (1.f - b1) * gi);
        float vi = fmaf(b2, v[i], (1.f - b2) * gi * gi);
        m[i] = mi;
        v[i] = vi;
        float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = pi - (lr / bc1) * (mi / denom);
    }
}

}  // namespace mri3d

using namespace mri3d;

static inline bool al16(const void* a, const void* b, const void* c = nullptr) {
    return ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c)) & 15) == 0;
}

extern "C" int mri3d_copy_channels(const void* src, void* dst, int64_t nvox, int32_t c, int32_t src_ld, int32_t dst_ld,
                                   int32_t dtype, mri3d_stream_t stream) {
    MRI3D_REQUIRE(dtype == MRI3D_F32, MRI3D_ENOTSUP, "copy_channels: only MRI3D_F32 is implemented");
    MRI3D_REQUIRE(src && dst && nvox > 0 && c > 0 && src_ld >= c && dst_ld >= c, MRI3D_EINVAL, "copy_channels: bad arguments");
    hipStream_t s = static_cast<hipStream_t>(stream);
    bool v4 = c % 4 == 0 && src_ld % 4 == 0 && dst_ld % 4 == 0 && al16(src, dst);
    int grid = stream_grid(nvox * (c / (v4 ? 4 : 1)), 256);
    if (v4)
        hipLaunchKernelGGL(copy_channels_kernel<4>, dim3(grid), dim3(256), 0, s, (const float*)src, (float*)dst, nvox, c, src_ld, dst_ld);
    else
        hipLaunchKernelGGL(copy_channels_kernel<1>, dim3(grid), dim3(256), 0, s, (const float*)src, (float*)dst, nvox, c, src_ld, dst_ld);
    return check_launch("copy_channels");
}

extern "C" int mri3d_add_channels(const void* a, const void* b, void* dst, int64_t nvox, int32_t c, int32_t a_ld,
                                  int32_t b_ld, int32_t dst_ld, int32_t dtype, mri3d_stream_t stream) {
    MRI3D_REQUIRE(dtype == MRI3D_F32, MRI3D_ENOTSUP, "add_channels: only MRI3D_F32 is implemented");
    MRI3D_REQUIRE(a && b && dst && nvox > 0 && c > 0 && a_ld >= c && b_ld >= c && dst_ld >= c, MRI3D_EINVAL,
                  "add_channels: bad arguments");
    hipStream_t s = static_cast<hipStream_t>(stream);
    bool v4 = c % 4 == 0 && a_ld % 4 == 0 && b_ld % 4 == 0 && dst_ld % 4 == 0 && al16(a, b, dst);
    int grid = stream_grid(nvox * (c / (v4 ? 4 : 1)), 256);
    if (v4)
        hipLaunchKernelGGL(add_channels_kernel<4>, dim3(grid), dim3(256), 0, s, (const float*)a, (const float*)b, (float*)dst, nvox, c, a_ld, b_ld, dst_ld);
    else
        hipLaunchKernelGGL(add_channels_kernel<1>, dim3(grid), dim3(256), 0, s, (const float*)a, (const float*)b, (float*)dst, nvox, c, a_ld, b_ld, dst_ld);
    return check_launch("add_channels");
}

extern "C" int mri3d_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                               float beta2, float eps, float weight_decay, int32_t step, float grad_scale,
                               int32_t decoupled, mri3d_stream_t stream) {
    MRI3D_REQUIRE(p && g && m && v && n > 0 && step >= 1, MRI3D_EINVAL, "adam_step: bad arguments");
    hipStream_t s = static_cast<hipStream_t>(stream);
    float bc1 = 1.f - powf(beta1, (float)step);
    float bc2s = sqrtf(1.f - powf(beta2, (float)step));
    hipLaunchKernelGGL(adam_kernel, dim3(stream_grid(n, 256)), dim3(256), 0, s, p, g, m, v, n, lr, beta1, beta2, eps,
                       weight_decay, bc1, bc2s, grad_scale, decoupled);
    return check_launch("adam_step");
}
