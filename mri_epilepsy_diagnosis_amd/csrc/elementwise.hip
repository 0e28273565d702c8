// elementwise.hip — channel-slice copy / add (torch.cat and residual sums on NDHWC buffers with a voxel pitch) and
// the flat-buffer Adam/AdamW step.  Reference: torch.cat in unet.UNet's decoder and modified_3dunet.py:158-178;
// `out += residual` modified_3dunet.py:108; torch.optim.AdamW segmentation/routine.py:358; Adam classification/routine.py:271.
// All HBM-bound streaming kernels (16 B per lane when alignment allows).
#include "common.h"

namespace mri3d {

// TS -> TD copy (TS == TD: torch.cat slice copy; TS != TD: the fp32 <-> bf16 cast at the edge of a bf16 region)
template <typename TS, typename TD, int VEC>
__global__ void __launch_bounds__(256)
copy_channels_kernel(const TS* __restrict__ src, TD* __restrict__ dst, int64_t nvox, int C, int s_ld, int d_ld) {
    const int CV = C / VEC;
    const int64_t total = nvox * CV;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int cv = (int)(i % CV);
        int64_t v = i / CV;
        if (VEC == 4)
            stf4(dst + v * d_ld + cv * 4, ldf4(src + v * s_ld + cv * 4));
        else
            stf(dst + v * d_ld + cv, ldf(src + v * s_ld + cv));
    }
}

template <typename T, int VEC>
__global__ void __launch_bounds__(256)
add_channels_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ dst, int64_t nvox,
                    int C, int a_ld, int b_ld, int d_ld) {
    const int CV = C / VEC;
    const int64_t total = nvox * CV;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int cv = (int)(i % CV);
        int64_t v = i / CV;
        if (VEC == 4) {
            float4 x = ldf4(a + v * a_ld + cv * 4);
            float4 y = ldf4(b + v * b_ld + cv * 4);
            stf4(dst + v * d_ld + cv * 4, make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w));
        } else {
            stf(dst + v * d_ld + cv, ldf(a + v * a_ld + cv) + ldf(b + v * b_ld + cv));
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

template <typename TS, typename TD>
static void launch_copy(const void* src, void* dst, int64_t nvox, int c, int src_ld, int dst_ld, bool v4, hipStream_t s) {
    int grid = stream_grid(nvox * (c / (v4 ? 4 : 1)), 256);
    if (v4)
        hipLaunchKernelGGL((copy_channels_kernel<TS, TD, 4>), dim3(grid), dim3(256), 0, s, (const TS*)src, (TD*)dst, nvox, c,
                           src_ld, dst_ld);
    else
        hipLaunchKernelGGL((copy_channels_kernel<TS, TD, 1>), dim3(grid), dim3(256), 0, s, (const TS*)src, (TD*)dst, nvox, c,
                           src_ld, dst_ld);
}

static inline bool known_dtype(int d) { return d == MRI3D_F32 || d == MRI3D_BF16; }

extern "C" int mri3d_copy_channels(const void* src, void* dst, int64_t nvox, int32_t c, int32_t src_ld, int32_t dst_ld,
                                   int32_t dtype, mri3d_stream_t stream) {
    return mri3d_convert_channels(src, dtype, dst, dtype, nvox, c, src_ld, dst_ld, stream);
}

extern "C" int mri3d_convert_channels(const void* src, int32_t src_dtype, void* dst, int32_t dst_dtype, int64_t nvox,
                                      int32_t c, int32_t src_ld, int32_t dst_ld, mri3d_stream_t stream) {
    MRI3D_REQUIRE(known_dtype(src_dtype) && known_dtype(dst_dtype), MRI3D_ENOTSUP, "convert_channels: unknown dtype");
    MRI3D_REQUIRE(src && dst && nvox > 0 && c > 0 && src_ld >= c && dst_ld >= c, MRI3D_EINVAL, "convert_channels: bad arguments");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool v4 = c % 4 == 0 && src_ld % 4 == 0 && dst_ld % 4 == 0 && aligned_vec4(src_dtype, src) && aligned_vec4(dst_dtype, dst);
    if (src_dtype == MRI3D_F32 && dst_dtype == MRI3D_F32) launch_copy<float, float>(src, dst, nvox, c, src_ld, dst_ld, v4, s);
    else if (src_dtype == MRI3D_F32) launch_copy<float, bf16_t>(src, dst, nvox, c, src_ld, dst_ld, v4, s);
    else if (dst_dtype == MRI3D_F32) launch_copy<bf16_t, float>(src, dst, nvox, c, src_ld, dst_ld, v4, s);
    else launch_copy<bf16_t, bf16_t>(src, dst, nvox, c, src_ld, dst_ld, v4, s);
    return check_launch("convert_channels");
}

extern "C" int mri3d_add_channels(const void* a, const void* b, void* dst, int64_t nvox, int32_t c, int32_t a_ld,
                                  int32_t b_ld, int32_t dst_ld, int32_t dtype, mri3d_stream_t stream) {
    MRI3D_REQUIRE(known_dtype(dtype), MRI3D_ENOTSUP, "add_channels: unknown dtype %d", dtype);
    MRI3D_REQUIRE(a && b && dst && nvox > 0 && c > 0 && a_ld >= c && b_ld >= c && dst_ld >= c, MRI3D_EINVAL,
                  "add_channels: bad arguments");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool v4 = c % 4 == 0 && a_ld % 4 == 0 && b_ld % 4 == 0 && dst_ld % 4 == 0 && aligned_vec4(dtype, a, b, dst);
    const int grid = stream_grid(nvox * (c / (v4 ? 4 : 1)), 256);
    MRI3D_DISPATCH_DTYPE(dtype, T, {
        if (v4)
            hipLaunchKernelGGL((add_channels_kernel<T, 4>), dim3(grid), dim3(256), 0, s, (const T*)a, (const T*)b, (T*)dst, nvox,
                               c, a_ld, b_ld, dst_ld);
        else
            hipLaunchKernelGGL((add_channels_kernel<T, 1>), dim3(grid), dim3(256), 0, s, (const T*)a, (const T*)b, (T*)dst, nvox,
                               c, a_ld, b_ld, dst_ld);
    });
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
