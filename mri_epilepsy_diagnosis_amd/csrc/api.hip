// api.hip — C-ABI entry points that are not tied to one kernel file: version, error string, and the Conv3d
// dispatcher (generic direct kernels vs. the MFMA implicit-GEMM path for 3x3x3 stride-1 layers).
#include "common.h"
#include <string.h>

namespace mri3d {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// conv_generic.hip
size_t conv_generic_workspace_bytes(const Mri3dConvGeom& g, int pass);
// activations (x, y, dy, dx) are in the storage type g.dtype; weights, bias and their gradients are fp32
int conv_generic_fwd(const Mri3dConvGeom& g, const void* x, const float* w, const float* bias, void* y, void* ws,
                     size_t ws_bytes, hipStream_t s);
int conv_generic_dgrad(const Mri3dConvGeom& g, const void* dy, const float* w, const float* bias, void* dx, void* ws,
                       size_t ws_bytes, hipStream_t s);
int conv_generic_wgrad(const Mri3dConvGeom& g, const void* x, const void* dy, float* dw, float* dbias, void* ws,
                       size_t ws_bytes, hipStream_t s);
// conv_mfma.hip
bool conv_mfma_supported(const Mri3dConvGeom& g, int pass);
size_t conv_mfma_workspace_bytes(const Mri3dConvGeom& g, int pass);
int conv_mfma_fwd(const Mri3dConvGeom& g, const void* x, const float* w, const float* bias, void* y, void* ws,
                  size_t ws_bytes, hipStream_t s);
int conv_mfma_dgrad(const Mri3dConvGeom& g, const void* dy, const float* w, const float* bias, void* dx, void* ws,
                    size_t ws_bytes, hipStream_t s);
int conv_mfma_wgrad(const Mri3dConvGeom& g, const void* x, const void* dy, float* dw, float* dbias, void* ws,
                    size_t ws_bytes, hipStream_t s);
int conv_mfma_fwd_stat_blocks(const Mri3dConvGeom& g);
bool conv_mfma_cat_supported(const Mri3dConvGeom& g, int split, int second_ld, int pass);
int conv_mfma_fwd_cat(const Mri3dConvGeom& g, const void* x, const void* x2, int split, int x2_ld, const float* w, const float* bias,
                      void* y, double* stat_part, void* ws, size_t ws_bytes, hipStream_t s);
int conv_mfma_dgrad_cat(const Mri3dConvGeom& g, const void* dy, const float* w, void* dx, void* dx2, int split, int dx2_ld, void* ws,
                        size_t ws_bytes, hipStream_t s);
int conv_mfma_wgrad_cat(const Mri3dConvGeom& g, const void* x, const void* x2, int split, int x2_ld, const void* dy, float* dw,
                        float* dbias, void* ws, size_t ws_bytes, hipStream_t s);
int conv_mfma_fwd_stats(const Mri3dConvGeom& g, const void* x, const float* w, const float* bias, void* y, double* stat_part,
                        void* ws, size_t ws_bytes, hipStream_t s);

// conv_march.hip
bool conv_march_takes(const Mri3dConvGeom& g, bool dgrad, bool stats, bool force);
int conv_march_stat_blocks(const Mri3dConvGeom& g, bool force);
int conv_march_run(const Mri3dConvGeom& g, bool dgrad, bool force, const void* in_v, const float* w, const float* bias, void* out_v,
                   void* ws, size_t ws_bytes, hipStream_t s, double* stat_part, const void* second, int split, int second_ld);

// conv_pointwise.hip
bool conv_pointwise_supported(const Mri3dConvGeom& g, int pass);
size_t conv_pointwise_workspace_bytes(const Mri3dConvGeom& g, int pass);
int conv_pointwise_fwd(const Mri3dConvGeom& g, const void* x, const float* w, const float* bias, void* y, hipStream_t s);
int conv_pointwise_dgrad(const Mri3dConvGeom& g, const void* dy, const float* w, const float* bias, void* dx,
                         hipStream_t s);
int conv_pointwise_wgrad(const Mri3dConvGeom& g, const void* x, const void* dy, float* dw, float* dbias, void* ws,
                         size_t ws_bytes, hipStream_t s);

// the MFMA kernels move 16-byte pieces: a pitched channel slice whose base is not 16-byte aligned (legal for the generic
// kernels) must not be routed to them

// first_ci >= 0: the input is split over two tensors (the *_cat entry points) and x_ld is the pitch of the first, which holds
// first_ci channels
static int conv_check(const Mri3dConvGeom* g, const char* who, int first_ci = -1) {
    MRI3D_REQUIRE(g != nullptr, MRI3D_EINVAL, "%s: null geometry", who);
    MRI3D_REQUIRE(g->dtype == MRI3D_F32 || g->dtype == MRI3D_BF16, MRI3D_ENOTSUP, "%s: unknown dtype %d", who, g->dtype);
    MRI3D_REQUIRE(g->n > 0 && g->di > 0 && g->hi > 0 && g->wi > 0 && g->ci > 0 && g->dout > 0 && g->ho > 0 && g->wo > 0 &&
                      g->co > 0,
                  MRI3D_EINVAL, "%s: empty tensor", who);
    MRI3D_REQUIRE(g->kd > 0 && g->kh > 0 && g->kw > 0 && g->sd > 0 && g->sh > 0 && g->sw > 0 && g->dd > 0 && g->dh > 0 &&
                      g->dw > 0 && g->pd >= 0 && g->ph >= 0 && g->pw >= 0,
                  MRI3D_EINVAL, "%s: bad kernel/stride/padding/dilation", who);
    MRI3D_REQUIRE(g->x_ld >= (first_ci >= 0 ? first_ci : g->ci) && g->y_ld >= g->co, MRI3D_EINVAL, "%s: pitch smaller than channel count", who);
    // torch: out = floor((in + 2p - d(k-1) - 1)/s) + 1
    int ed = (g->di + 2 * g->pd - g->dd * (g->kd - 1) - 1) / g->sd + 1;
    int eh = (g->hi + 2 * g->ph - g->dh * (g->kh - 1) - 1) / g->sh + 1;
    int ew = (g->wi + 2 * g->pw - g->dw * (g->kw - 1) - 1) / g->sw + 1;
    MRI3D_REQUIRE(ed == g->dout && eh == g->ho && ew == g->wo, MRI3D_EINVAL,
                  "%s: output dims (%d,%d,%d) do not match conv arithmetic (%d,%d,%d)", who, g->dout, g->ho, g->wo, ed, eh,
                  ew);
    return MRI3D_OK;
}

}  // namespace mri3d

using namespace mri3d;

extern "C" int mri3d_version(void) { return 100; }  // 0.1.0
extern "C" const char* mri3d_last_error(void) { return g_err; }

extern "C" size_t mri3d_conv3d_workspace_bytes(const Mri3dConvGeom* g, int pass) {
    if (!g) return 0;
    size_t a = conv_generic_workspace_bytes(*g, pass);
    size_t b = conv_mfma_supported(*g, pass) ? conv_mfma_workspace_bytes(*g, pass) : 0;
    size_t c = conv_pointwise_supported(*g, pass) ? conv_pointwise_workspace_bytes(*g, pass) : 0;
    a = a > b ? a : b;
    return align_up(a > c ? a : c, 256);
}

extern "C" int mri3d_conv3d_fwd(const Mri3dConvGeom* g, const void* x, const void* w, const void* bias, void* y,
                                void* workspace, size_t ws_bytes, mri3d_stream_t stream) {
    int rc = conv_check(g, "conv3d_fwd");
    if (rc) return rc;
    MRI3D_REQUIRE(x && w && y, MRI3D_EINVAL, "conv3d_fwd: null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (conv_mfma_supported(*g, MRI3D_PASS_FWD) && aligned16(x, y, workspace))
        return conv_mfma_fwd(*g, x, (const float*)w, (const float*)bias, y, workspace, ws_bytes, s);
    if (conv_pointwise_supported(*g, MRI3D_PASS_FWD) && aligned_vec4(g->dtype, x))
        return conv_pointwise_fwd(*g, x, (const float*)w, (const float*)bias, y, s);
    return conv_generic_fwd(*g, x, (const float*)w, (const float*)bias, y, workspace, ws_bytes, s);
}

extern "C" int32_t mri3d_conv3d_fwd_stats_blocks(const Mri3dConvGeom* g) {
    if (!g || conv_check(g, "conv3d_fwd_stats_blocks") != MRI3D_OK) return 0;
    return conv_mfma_supported(*g, MRI3D_PASS_FWD) ? conv_mfma_fwd_stat_blocks(*g) : 0;
}

extern "C" int mri3d_conv3d_fwd_stats(const Mri3dConvGeom* g, const void* x, const void* w, const void* bias, void* y,
                                      double* stat_partials, void* workspace, size_t ws_bytes, mri3d_stream_t stream) {
    int rc = conv_check(g, "conv3d_fwd_stats");
    if (rc) return rc;
    MRI3D_REQUIRE(x && w && y && stat_partials, MRI3D_EINVAL, "conv3d_fwd_stats: null pointer");
    MRI3D_REQUIRE(conv_mfma_supported(*g, MRI3D_PASS_FWD) && conv_mfma_fwd_stat_blocks(*g) > 0 && aligned16(x, y, workspace),
                  MRI3D_ENOTSUP, "conv3d_fwd_stats: geometry not served by the MFMA forward kernel (query mri3d_conv3d_fwd_stats_blocks)");
    return conv_mfma_fwd_stats(*g, x, (const float*)w, (const float*)bias, y, stat_partials, workspace, ws_bytes,
                               static_cast<hipStream_t>(stream));
}

extern "C" int mri3d_conv3d_dgrad(const Mri3dConvGeom* g, const void* dy, const void* w, const void* bias, void* dx,
                                  void* workspace, size_t ws_bytes, mri3d_stream_t stream) {
    int rc = conv_check(g, "conv3d_dgrad");
    if (rc) return rc;
    MRI3D_REQUIRE(dy && w && dx, MRI3D_EINVAL, "conv3d_dgrad: null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (conv_pointwise_supported(*g, MRI3D_PASS_DGRAD) && aligned_vec4(g->dtype, dx))
        return conv_pointwise_dgrad(*g, dy, (const float*)w, (const float*)bias, dx, s);
    if (conv_mfma_supported(*g, MRI3D_PASS_DGRAD) && aligned16(dy, dx, workspace))
        return conv_mfma_dgrad(*g, dy, (const float*)w, (const float*)bias, dx, workspace, ws_bytes, s);
    return conv_generic_dgrad(*g, dy, (const float*)w, (const float*)bias, dx, workspace, ws_bytes, s);
}

// ---- convolution over torch.cat((x, x2), dim=1) without the concatenation (unet.UNet decoder: cat((skip, upsampled)))
extern "C" int32_t mri3d_conv3d_cat_supported(const Mri3dConvGeom* g, int32_t split, int32_t second_ld, int32_t pass) {
    if (!g || split <= 0 || conv_check(g, "conv3d_cat_supported", split) != MRI3D_OK) return 0;
    return conv_mfma_cat_supported(*g, split, second_ld, pass) ? 1 : 0;
}

extern "C" int mri3d_conv3d_fwd_cat(const Mri3dConvGeom* g, const void* x, const void* x2, int32_t split, int32_t x2_ld,
                                    const void* w, const void* bias, void* y, double* stat_partials, void* workspace,
                                    size_t ws_bytes, mri3d_stream_t stream) {
    int rc = conv_check(g, "conv3d_fwd_cat", split);
    if (rc) return rc;
    MRI3D_REQUIRE(x && x2 && w && y, MRI3D_EINVAL, "conv3d_fwd_cat: null pointer");
    MRI3D_REQUIRE(conv_mfma_cat_supported(*g, split, x2_ld, MRI3D_PASS_FWD) && aligned16(x, y, workspace) && aligned16(x2),
                  MRI3D_ENOTSUP, "conv3d_fwd_cat: geometry / alignment not served (query mri3d_conv3d_cat_supported)");
    MRI3D_REQUIRE(stat_partials == nullptr || conv_mfma_fwd_stat_blocks(*g) > 0, MRI3D_ENOTSUP, "conv3d_fwd_cat: no fused statistics for this geometry");
    return conv_mfma_fwd_cat(*g, x, x2, split, x2_ld, (const float*)w, (const float*)bias, y, stat_partials, workspace, ws_bytes,
                             static_cast<hipStream_t>(stream));
}

extern "C" int mri3d_conv3d_dgrad_cat(const Mri3dConvGeom* g, const void* dy, const void* w, void* dx, void* dx2, int32_t split,
                                      int32_t dx2_ld, void* workspace, size_t ws_bytes, mri3d_stream_t stream) {
    int rc = conv_check(g, "conv3d_dgrad_cat", split);
    if (rc) return rc;
    MRI3D_REQUIRE(dy && w && dx && dx2, MRI3D_EINVAL, "conv3d_dgrad_cat: null pointer");
    MRI3D_REQUIRE(conv_mfma_cat_supported(*g, split, dx2_ld, MRI3D_PASS_DGRAD) && aligned16(dy, dx, workspace) && aligned16(dx2),
                  MRI3D_ENOTSUP, "conv3d_dgrad_cat: geometry / alignment not served (query mri3d_conv3d_cat_supported)");
    return conv_mfma_dgrad_cat(*g, dy, (const float*)w, dx, dx2, split, dx2_ld, workspace, ws_bytes, static_cast<hipStream_t>(stream));
}

extern "C" int mri3d_conv3d_wgrad_cat(const Mri3dConvGeom* g, const void* x, const void* x2, int32_t split, int32_t x2_ld,
                                      const void* dy, void* dw, void* dbias, void* workspace, size_t ws_bytes,
                                      mri3d_stream_t stream) {
    int rc = conv_check(g, "conv3d_wgrad_cat", split);
    if (rc) return rc;
    MRI3D_REQUIRE(x && x2 && dy && dw, MRI3D_EINVAL, "conv3d_wgrad_cat: null pointer");
    MRI3D_REQUIRE(conv_mfma_cat_supported(*g, split, x2_ld, MRI3D_PASS_WGRAD) && aligned16(x, dy, workspace) && aligned16(x2),
                  MRI3D_ENOTSUP, "conv3d_wgrad_cat: geometry / alignment not served (query mri3d_conv3d_cat_supported)");
    return conv_mfma_wgrad_cat(*g, x, x2, split, x2_ld, dy, (float*)dw, (float*)dbias, workspace, ws_bytes,
                               static_cast<hipStream_t>(stream));
}

// ---- the d-marching forward / data-gradient kernel by name (conv_march.hip).  The plain entry points above choose it themselves for
// the layers it is faster on; these take EVERY geometry it can compute, so that parity tests reach it with small volumes.
extern "C" int32_t mri3d_conv3d_march_supported(const Mri3dConvGeom* g, int32_t pass) {
    if (!g || (pass != MRI3D_PASS_FWD && pass != MRI3D_PASS_DGRAD) || conv_check(g, "conv3d_march_supported") != MRI3D_OK) return 0;
    return conv_march_takes(*g, pass == MRI3D_PASS_DGRAD, false, true) ? 1 : 0;
}

extern "C" int32_t mri3d_conv3d_march_stats_blocks(const Mri3dConvGeom* g) {
    if (!g || conv_check(g, "conv3d_march_stats_blocks") != MRI3D_OK) return 0;
    return conv_march_stat_blocks(*g, true);
}

extern "C" int mri3d_conv3d_fwd_march(const Mri3dConvGeom* g, const void* x, const void* x2, int32_t split, int32_t x2_ld,
                                      const void* w, const void* bias, void* y, double* stat_partials, void* workspace,
                                      size_t ws_bytes, mri3d_stream_t stream) {
    int rc = conv_check(g, "conv3d_fwd_march", x2 ? split : -1);
    if (rc) return rc;
    MRI3D_REQUIRE(x && w && y, MRI3D_EINVAL, "conv3d_fwd_march: null pointer");
    MRI3D_REQUIRE(x2 == nullptr || (split > 0 && split < g->ci && x2_ld >= g->ci - split), MRI3D_EINVAL, "conv3d_fwd_march: bad split");
    return conv_march_run(*g, false, true, x, (const float*)w, (const float*)bias, y, workspace, ws_bytes,
                          static_cast<hipStream_t>(stream), stat_partials, x2, split, x2_ld);
}

extern "C" int mri3d_conv3d_dgrad_march(const Mri3dConvGeom* g, const void* dy, const void* w, void* dx, void* dx2, int32_t split,
                                        int32_t dx2_ld, void* workspace, size_t ws_bytes, mri3d_stream_t stream) {
    int rc = conv_check(g, "conv3d_dgrad_march", dx2 ? split : -1);
    if (rc) return rc;
    MRI3D_REQUIRE(dy && w && dx, MRI3D_EINVAL, "conv3d_dgrad_march: null pointer");
    MRI3D_REQUIRE(dx2 == nullptr || (split > 0 && split < g->ci && dx2_ld >= g->ci - split), MRI3D_EINVAL, "conv3d_dgrad_march: bad split");
    return conv_march_run(*g, true, true, dy, (const float*)w, nullptr, dx, workspace, ws_bytes, static_cast<hipStream_t>(stream),
                          nullptr, dx2, split, dx2_ld);
}

extern "C" int mri3d_conv3d_wgrad(const Mri3dConvGeom* g, const void* x, const void* dy, void* dw, void* dbias,
                                  void* workspace, size_t ws_bytes, mri3d_stream_t stream) {
    int rc = conv_check(g, "conv3d_wgrad");
    if (rc) return rc;
    MRI3D_REQUIRE(x && dy && dw, MRI3D_EINVAL, "conv3d_wgrad: null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (conv_pointwise_supported(*g, MRI3D_PASS_WGRAD) && aligned_vec4(g->dtype, x))
        return conv_pointwise_wgrad(*g, x, dy, (float*)dw, (float*)dbias, workspace, ws_bytes, s);
    if (conv_mfma_supported(*g, MRI3D_PASS_WGRAD) && aligned16(x, dy, workspace))
        return conv_mfma_wgrad(*g, x, dy, (float*)dw, (float*)dbias, workspace, ws_bytes, s);
    return conv_generic_wgrad(*g, x, dy, (float*)dw, (float*)dbias, workspace, ws_bytes, s);
}
