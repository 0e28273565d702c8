// patches.hip — the patch pipeline either side of the training path: cutting 64^3 training patches out
// of HBM-resident volumes and writing the predicted label patches of a grid inference back into the whole volume.
// Reference call sites: torchio.Queue(... sampler_class=torchio.sampler.ImageSampler ...) segmentation/routine.py:150-178,
// segmentation/pretraining_3d_unet.ipynb cell 24; torchio.inference.GridSampler / GridAggregator.add_batch(labels, locations)
// with labels = logits.argmax(dim=1, keepdim=True), pretraining_3d_unet.ipynb cell 26.  TorchIO is a third-party dependency
// that is absent from the reference tree and from this image ("parity unpinned"); its window arithmetic is restated in
// oracle/patches.py and mirrored here:
//   extract:    out[p, z, y, x] = vol[loc[p].vol, loc[p].d0 + z, loc[p].h0 + y, loc[p].w0 + x]      (any element size)
//   aggregate:  every window is cropped by `border` voxels on all six faces and written to the output volume in patch
//               order, so where two cropped windows overlap (the extra last window of an axis) the LATER patch wins.
// Both are pure HBM copies (<= 1 read + 1 write per element).  Patch origins arrive as a HOST table: the library checks
// them against the volume (an out-of-range origin would otherwise be an out-of-bounds access) and hands them to the kernel
// by value, 64 windows per launch, so there is no device-side table, no H2D copy and the launches can be graph-captured.
// The "later patch wins" rule is resolved without ordering the writes: a lane writes voxel v of window p only when no later
// window of the same launch also covers v; successive launches are ordered by the stream.
#include "common.h"
#include <algorithm>

namespace mri3d {

constexpr int kPatchChunk = 64;

struct PatchTable {
    int n;
    int vol[kPatchChunk];
    int d0[kPatchChunk], h0[kPatchChunk], w0[kPatchChunk];
};

template <typename E>
__global__ void __launch_bounds__(256)
extract_patches_kernel(const E* __restrict__ vol, E* __restrict__ out, PatchTable t, int d, int h, int w, int pd, int ph,
                       int pw) {
    const int p = blockIdx.y;
    const unsigned per = (unsigned)pd * ph * pw;
    const E* src = vol + (((long long)t.vol[p] * d + t.d0[p]) * h + t.h0[p]) * (long long)w + t.w0[p];
    E* dst = out + (long long)p * per;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < per; i += gridDim.x * blockDim.x) {
        const unsigned x = i % pw, r = i / pw;
        const unsigned y = r % ph, z = r / ph;
        dst[i] = src[((long long)z * h + y) * w + x];
    }
}

template <typename T>
__device__ __forceinline__ uint8_t argmax_c(const T* z, int C) {
    float best = ldf(z);
    int bi = 0;
    for (int j = 1; j < C; ++j) {
        float t = ldf(z + j);
        // torch.argmax: first maximal value; NaN counts as maximal (same rule as argmax_u8_kernel, loss.hip)
        if ((t > best && best == best) || (t != t && best == best)) { best = t; bi = j; }
    }
    return (uint8_t)bi;
}

// SRC = uint8_t: label patches [P][pd][ph][pw];  SRC = float / bf16_t: logits [P][pd][ph][pw][ld], arg-max taken here.
template <typename SRC>
__global__ void __launch_bounds__(256)
aggregate_patches_kernel(const SRC* __restrict__ patches, uint8_t* __restrict__ out, PatchTable t, int d, int h, int w,
                         int pd, int ph, int pw, int bd, int bh, int bw, int C, int ld) {
    const int p = blockIdx.y;
    const int cd = pd - 2 * bd, chh = ph - 2 * bh, cw = pw - 2 * bw;  // cropped window
    const unsigned per = (unsigned)cd * chh * cw;
    const int od = t.d0[p] + bd, oh = t.h0[p] + bh, ow = t.w0[p] + bw;  // its origin in the volume
    const int vol = t.vol[p];
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < per; i += gridDim.x * blockDim.x) {
        const unsigned x = i % cw, r = i / cw;
        const unsigned y = r % chh, z = r / chh;
        const int gz = od + (int)z, gy = oh + (int)y, gx = ow + (int)x;
        bool owner = true;
        for (int q = p + 1; q < t.n; ++q) {
            const bool covered = t.vol[q] == vol && (unsigned)(gz - t.d0[q] - bd) < (unsigned)cd &&
                                 (unsigned)(gy - t.h0[q] - bh) < (unsigned)chh && (unsigned)(gx - t.w0[q] - bw) < (unsigned)cw;
            if (covered) { owner = false; break; }
        }
        if (!owner) continue;
        const long long s = (((long long)p * pd + (z + bd)) * ph + (y + bh)) * pw + (x + bw);
        uint8_t v;
        if constexpr (sizeof(SRC) == 1) v = patches[s];
        else v = argmax_c(patches + s * ld, C);
        out[(((long long)vol * d + gz) * h + gy) * (long long)w + gx] = v;
    }
}

static int fill_table(PatchTable& t, const int32_t* loc, int first, int count, int nvol, int d, int h, int w, int pd, int ph,
                      int pw, const char* who) {
    t.n = count;
    for (int i = 0; i < count; ++i) {
        const int32_t* l = loc + 4 * (size_t)(first + i);
        MRI3D_REQUIRE(l[0] >= 0 && l[0] < nvol && l[1] >= 0 && l[2] >= 0 && l[3] >= 0 && (int64_t)l[1] + pd <= d &&
                          (int64_t)l[2] + ph <= h && (int64_t)l[3] + pw <= w,
                      MRI3D_EINVAL, "%s: window %d = (vol %d, origin %d,%d,%d, size %d,%d,%d) leaves the %d x (%d,%d,%d) volumes",
                      who, first + i, l[0], l[1], l[2], l[3], pd, ph, pw, nvol, d, h, w);
        t.vol[i] = l[0];
        t.d0[i] = l[1];
        t.h0[i] = l[2];
        t.w0[i] = l[3];
    }
    for (int i = count; i < kPatchChunk; ++i) t.vol[i] = t.d0[i] = t.h0[i] = t.w0[i] = 0;
    return MRI3D_OK;
}

}  // namespace mri3d

using namespace mri3d;

extern "C" int mri3d_extract_patches(const void* volumes, int32_t elem_bytes, int32_t nvol, int32_t d, int32_t h, int32_t w,
                                     const int32_t* loc_host, int32_t npatch, int32_t pd, int32_t ph, int32_t pw, void* out,
                                     mri3d_stream_t stream) {
    MRI3D_REQUIRE(volumes && out && loc_host && nvol > 0 && d > 0 && h > 0 && w > 0 && npatch > 0 && pd > 0 && ph > 0 && pw > 0,
                  MRI3D_EINVAL, "extract_patches: bad arguments");
    MRI3D_REQUIRE(elem_bytes == 1 || elem_bytes == 2 || elem_bytes == 4 || elem_bytes == 8 || elem_bytes == 16, MRI3D_ENOTSUP,
                  "extract_patches: element size %d (1, 2, 4, 8 or 16 bytes)", elem_bytes);
    MRI3D_REQUIRE(((reinterpret_cast<uintptr_t>(volumes) | reinterpret_cast<uintptr_t>(out)) & (uintptr_t)(elem_bytes - 1)) == 0,
                  MRI3D_EINVAL, "extract_patches: pointers not aligned to the element size");
    MRI3D_REQUIRE((int64_t)pd * ph * pw < 0x7fffffffLL, MRI3D_ENOTSUP, "extract_patches: patch too large");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int64_t per = (int64_t)pd * ph * pw;
    const int gx = (int)std::min<int64_t>(cdiv64(per, 256 * 4), 4096);
    PatchTable t;
    for (int first = 0; first < npatch; first += kPatchChunk) {
        const int count = std::min(kPatchChunk, npatch - first);
        int rc = fill_table(t, loc_host, first, count, nvol, d, h, w, pd, ph, pw, "extract_patches");
        if (rc) return rc;
        char* o = static_cast<char*>(out) + (size_t)first * per * elem_bytes;
        dim3 grid(gx, count);
        switch (elem_bytes) {
            case 1:
                hipLaunchKernelGGL(extract_patches_kernel<uint8_t>, grid, dim3(256), 0, s, (const uint8_t*)volumes, (uint8_t*)o,
                                   t, d, h, w, pd, ph, pw);
                break;
            case 2:
                hipLaunchKernelGGL(extract_patches_kernel<uint16_t>, grid, dim3(256), 0, s, (const uint16_t*)volumes,
                                   (uint16_t*)o, t, d, h, w, pd, ph, pw);
                break;
            case 4:
                hipLaunchKernelGGL(extract_patches_kernel<uint32_t>, grid, dim3(256), 0, s, (const uint32_t*)volumes,
                                   (uint32_t*)o, t, d, h, w, pd, ph, pw);
                break;
            case 8:
                hipLaunchKernelGGL(extract_patches_kernel<uint2>, grid, dim3(256), 0, s, (const uint2*)volumes, (uint2*)o, t, d,
                                   h, w, pd, ph, pw);
                break;
            default:
                hipLaunchKernelGGL(extract_patches_kernel<uint4>, grid, dim3(256), 0, s, (const uint4*)volumes, (uint4*)o, t, d,
                                   h, w, pd, ph, pw);
                break;
        }
    }
    return check_launch("extract_patches");
}

static int aggregate_common(const void* src, int src_kind, int32_t c, int32_t ld, const int32_t* loc_host, int32_t npatch,
                            int32_t pd, int32_t ph, int32_t pw, int32_t bd, int32_t bh, int32_t bw, uint8_t* out, int32_t nvol,
                            int32_t d, int32_t h, int32_t w, mri3d_stream_t stream, const char* who) {
    MRI3D_REQUIRE(src && out && loc_host && nvol > 0 && d > 0 && h > 0 && w > 0 && npatch > 0 && pd > 0 && ph > 0 && pw > 0,
                  MRI3D_EINVAL, "%s: bad arguments", who);
    MRI3D_REQUIRE(bd >= 0 && bh >= 0 && bw >= 0 && 2 * bd < pd && 2 * bh < ph && 2 * bw < pw, MRI3D_EINVAL,
                  "%s: border (%d,%d,%d) leaves nothing of a (%d,%d,%d) window", who, bd, bh, bw, pd, ph, pw);
    MRI3D_REQUIRE((int64_t)pd * ph * pw < 0x7fffffffLL, MRI3D_ENOTSUP, "%s: patch too large", who);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int64_t per = (int64_t)(pd - 2 * bd) * (ph - 2 * bh) * (pw - 2 * bw);
    const int gx = (int)std::min<int64_t>(cdiv64(per, 256 * 4), 4096);
    const int64_t per_full = (int64_t)pd * ph * pw;
    PatchTable t;
    for (int first = 0; first < npatch; first += kPatchChunk) {
        const int count = std::min(kPatchChunk, npatch - first);
        int rc = fill_table(t, loc_host, first, count, nvol, d, h, w, pd, ph, pw, who);
        if (rc) return rc;
        dim3 grid(gx, count);
        if (src_kind == 0) {
            hipLaunchKernelGGL(aggregate_patches_kernel<uint8_t>, grid, dim3(256), 0, s,
                               (const uint8_t*)src + (size_t)first * per_full, out, t, d, h, w, pd, ph, pw, bd, bh, bw, 1, 1);
        } else if (src_kind == 1) {
            hipLaunchKernelGGL(aggregate_patches_kernel<float>, grid, dim3(256), 0, s,
                               (const float*)src + (size_t)first * per_full * ld, out, t, d, h, w, pd, ph, pw, bd, bh, bw, c, ld);
        } else {
            hipLaunchKernelGGL(aggregate_patches_kernel<bf16_t>, grid, dim3(256), 0, s,
                               (const bf16_t*)src + (size_t)first * per_full * ld, out, t, d, h, w, pd, ph, pw, bd, bh, bw, c, ld);
        }
    }
    return check_launch(who);
}

extern "C" int mri3d_aggregate_patches_u8(const uint8_t* patches, const int32_t* loc_host, int32_t npatch, int32_t pd,
                                          int32_t ph, int32_t pw, int32_t bd, int32_t bh, int32_t bw, uint8_t* out,
                                          int32_t nvol, int32_t d, int32_t h, int32_t w, mri3d_stream_t stream) {
    return aggregate_common(patches, 0, 1, 1, loc_host, npatch, pd, ph, pw, bd, bh, bw, out, nvol, d, h, w, stream,
                            "aggregate_patches_u8");
}

extern "C" int mri3d_aggregate_patches_argmax(const void* logits, int32_t c, int32_t ld, int32_t dtype,
                                              const int32_t* loc_host, int32_t npatch, int32_t pd, int32_t ph, int32_t pw,
                                              int32_t bd, int32_t bh, int32_t bw, uint8_t* out, int32_t nvol, int32_t d,
                                              int32_t h, int32_t w, mri3d_stream_t stream) {
    MRI3D_REQUIRE(dtype == MRI3D_F32 || dtype == MRI3D_BF16, MRI3D_ENOTSUP, "aggregate_patches_argmax: unknown dtype %d", dtype);
    MRI3D_REQUIRE(c > 0 && c <= 256 && ld >= c, MRI3D_EINVAL, "aggregate_patches_argmax: bad channel count / pitch");
    return aggregate_common(logits, dtype == MRI3D_F32 ? 1 : 2, c, ld, loc_host, npatch, pd, ph, pw, bd, bh, bw, out, nvol, d, h,
                            w, stream, "aggregate_patches_argmax");
}
