// Host-side sanitizer driver: the C-ABI library's HOST code — argument validation, plan /
// dispatch selection, workspace sizing, error-string formatting — compiled for the host only with
// -fsanitize=address,undefined together with this file, and driven WITHOUT a GPU: every call below must return before any
// kernel launch (invalid arguments, or a pure workspace query).  GPU AddressSanitizer is not available on this pool; the
// device code is covered by the parity suite instead.  Exit code 0 and no sanitizer report = pass.
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include "mri3d.h"

static int g_fail = 0;
#define EXPECT(cond)                                                        \
    do {                                                                    \
        if (!(cond)) {                                                      \
            fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); \
            ++g_fail;                                                       \
        }                                                                   \
    } while (0)

static Mri3dConvGeom conv(int n, int d, int h, int w, int ci, int co, int k, int s, int p, int dil, int dtype, int xpad,
                          int ypad) {
    Mri3dConvGeom g;
    memset(&g, 0, sizeof(g));
    g.n = n; g.di = d; g.hi = h; g.wi = w; g.ci = ci; g.co = co;
    g.kd = g.kh = g.kw = k; g.sd = g.sh = g.sw = s; g.pd = g.ph = g.pw = p; g.dd = g.dh = g.dw = dil;
    g.dout = (d + 2 * p - dil * (k - 1) - 1) / s + 1;
    g.ho = (h + 2 * p - dil * (k - 1) - 1) / s + 1;
    g.wo = (w + 2 * p - dil * (k - 1) - 1) / s + 1;
    g.x_ld = ci + xpad; g.y_ld = co + ypad; g.dtype = dtype;
    return g;
}

int main() {
    EXPECT(mri3d_version() >= 100);
    char fake[64];                       // a non-null pointer that is never dereferenced on the host
    void* P = fake;

    // ---- conv: workspace queries over a sweep of geometries (plan functions of every kernel family), incl. degenerate ones
    const int chans[] = {1, 2, 3, 4, 8, 12, 16, 24, 32, 48, 64, 96, 128, 192};
    const int sizes[][3] = {{1, 1, 1}, {2, 3, 5}, {8, 8, 8}, {10, 12, 10}, {32, 32, 32}, {80, 96, 80}, {160, 192, 160}, {192, 192, 192}};
    size_t checked = 0;
    for (int dtype = 0; dtype < 2; ++dtype)
        for (int ci : chans)
            for (int co : chans)
                for (auto& sz : sizes)
                    for (int cfg = 0; cfg < 5; ++cfg) {
                        const int k = cfg == 3 ? 1 : (cfg == 4 ? 4 : 3), s = cfg == 1 ? 2 : (cfg == 4 ? 2 : 1);
                        const int p = cfg == 3 ? 0 : 1, dil = cfg == 2 ? 3 : 1;
                        if (sz[0] + 2 * p - dil * (k - 1) - 1 < 0 || sz[1] + 2 * p - dil * (k - 1) - 1 < 0 || sz[2] + 2 * p - dil * (k - 1) - 1 < 0) continue;
                        for (int n : {1, 2, 512}) {
                            if ((int64_t)n * sz[0] * sz[1] * sz[2] * (ci > co ? ci : co) > (int64_t)1 << 34) continue;
                            Mri3dConvGeom g = conv(n, sz[0], sz[1], sz[2], ci, co, k, s, p, dil, dtype, (ci % 3) * 2, (co % 2) * 4);
                            for (int pass = 0; pass < 3; ++pass) {
                                const size_t ws = mri3d_conv3d_workspace_bytes(&g, pass);
                                EXPECT(ws % 256 == 0 && ws < ((size_t)1 << 40));
                                ++checked;
                            }
                        }
                    }
    EXPECT(checked > 10000);
    EXPECT(mri3d_conv3d_workspace_bytes(nullptr, 0) == 0);

    // ---- conv: argument validation (every branch returns before a launch)
    Mri3dConvGeom g = conv(2, 16, 16, 16, 16, 16, 3, 1, 1, 1, MRI3D_F32, 0, 0);
    EXPECT(mri3d_conv3d_fwd(nullptr, P, P, P, P, P, 0, nullptr) == MRI3D_EINVAL);
    EXPECT(strlen(mri3d_last_error()) > 0);
    EXPECT(mri3d_conv3d_fwd(&g, nullptr, P, P, P, P, 0, nullptr) == MRI3D_EINVAL);
    EXPECT(mri3d_conv3d_dgrad(&g, P, nullptr, P, P, P, 0, nullptr) == MRI3D_EINVAL);
    EXPECT(mri3d_conv3d_wgrad(&g, P, P, nullptr, P, P, 0, nullptr) == MRI3D_EINVAL);
    {   // workspace too small for the MFMA path (16-byte aligned fake pointers)
        alignas(16) static char buf[64];
        EXPECT(mri3d_conv3d_fwd(&g, buf, buf, buf, buf, buf, 16, nullptr) == MRI3D_EWORKSPACE);
        EXPECT(mri3d_conv3d_wgrad(&g, buf, buf, buf, buf, buf, 16, nullptr) == MRI3D_EWORKSPACE);
        EXPECT(mri3d_conv3d_dgrad(&g, buf, buf, buf, buf, nullptr, 0, nullptr) == MRI3D_EWORKSPACE);
    }
    Mri3dConvGeom bad = g;
    bad.dtype = 7;
    EXPECT(mri3d_conv3d_fwd(&bad, P, P, P, P, P, 0, nullptr) == MRI3D_ENOTSUP);
    bad = g; bad.dout = 15;
    EXPECT(mri3d_conv3d_fwd(&bad, P, P, P, P, P, 0, nullptr) == MRI3D_EINVAL);
    EXPECT(strstr(mri3d_last_error(), "do not match") != nullptr);
    bad = g; bad.x_ld = 8;
    EXPECT(mri3d_conv3d_wgrad(&bad, P, P, P, P, P, 0, nullptr) == MRI3D_EINVAL);
    bad = g; bad.n = 0;
    EXPECT(mri3d_conv3d_dgrad(&bad, P, P, P, P, P, 0, nullptr) == MRI3D_EINVAL);
    bad = g; bad.kd = 0;
    EXPECT(mri3d_conv3d_fwd(&bad, P, P, P, P, P, 0, nullptr) == MRI3D_EINVAL);
    bad = g; bad.pd = -1;
    EXPECT(mri3d_conv3d_fwd(&bad, P, P, P, P, P, 0, nullptr) == MRI3D_EINVAL);

    // ---- split operands (conv over cat((x, x2))): the support query over a sweep, refusals without a launch
    {
        int served = 0;
        for (int dtype = 0; dtype < 2; ++dtype)
            for (int ca : {8, 16, 32, 48})
                for (int cb : {8, 16, 24, 32, 64})
                    for (int sz : {6, 40, 160}) {
                        Mri3dConvGeom c = conv(2, sz, sz + 8, sz, ca + cb, 16, 3, 1, 1, 1, dtype, 0, 0);
                        c.x_ld = ca;
                        for (int pass = 0; pass < 3; ++pass) served += mri3d_conv3d_cat_supported(&c, ca, cb, pass);
                        EXPECT(mri3d_conv3d_cat_supported(&c, 0, cb, 0) == 0);          // no first part
                        EXPECT(mri3d_conv3d_cat_supported(&c, ca + cb, cb, 0) == 0);    // no second part
                    }
        EXPECT(served > 0);
        EXPECT(mri3d_conv3d_cat_supported(nullptr, 16, 32, 0) == 0);
        Mri3dConvGeom c = conv(2, 160, 192, 160, 48, 16, 3, 1, 1, 1, 0, 0, 0);
        c.x_ld = 16;
        EXPECT(mri3d_conv3d_cat_supported(&c, 16, 32, 0) == 1 && mri3d_conv3d_cat_supported(&c, 16, 32, 1) == 1 &&
               mri3d_conv3d_cat_supported(&c, 16, 32, 2) == 1);
        EXPECT(mri3d_conv3d_cat_supported(&c, 8, 40, 0) == 0);                         // split not a multiple of 16
        EXPECT(mri3d_conv3d_fwd_cat(&c, nullptr, P, 16, 32, P, P, P, nullptr, P, 0, nullptr) == MRI3D_EINVAL);
        EXPECT(mri3d_conv3d_fwd_cat(&c, P, nullptr, 16, 32, P, P, P, nullptr, P, 0, nullptr) == MRI3D_EINVAL);
        EXPECT(mri3d_conv3d_dgrad_cat(&c, P, P, P, nullptr, 16, 32, P, 0, nullptr) == MRI3D_EINVAL);
        EXPECT(mri3d_conv3d_wgrad_cat(&c, P, nullptr, 16, 32, P, P, P, P, 0, nullptr) == MRI3D_EINVAL);
        alignas(16) static char buf2[64];
        EXPECT(mri3d_conv3d_fwd_cat(&c, buf2, buf2, 8, 40, buf2, buf2, buf2, nullptr, buf2, 0, nullptr) == MRI3D_ENOTSUP);   // bad split
        EXPECT(mri3d_conv3d_fwd_cat(&c, buf2, buf2 + 4, 16, 32, buf2, buf2, buf2, nullptr, buf2, 0, nullptr) == MRI3D_ENOTSUP);   // misaligned x2
        EXPECT(mri3d_conv3d_fwd_cat(&c, buf2, buf2, 16, 32, buf2, buf2, buf2, nullptr, buf2, 16, nullptr) == MRI3D_EWORKSPACE);
        c.x_ld = 8;   // pitch of the first tensor smaller than its channel count
        EXPECT(mri3d_conv3d_fwd_cat(&c, buf2, buf2, 16, 32, buf2, buf2, buf2, nullptr, buf2, 0, nullptr) == MRI3D_EINVAL);
    }

    // ---- other families: null geometry / null pointers / zero sizes must be refused, workspace queries must not crash
    EXPECT(mri3d_norm_stats(nullptr, P, nullptr, nullptr, nullptr, nullptr, 0.1f, P, 0, nullptr) != MRI3D_OK);
    EXPECT(mri3d_norm_act_fwd(nullptr, P, nullptr, nullptr, nullptr, nullptr, nullptr, P, nullptr) != MRI3D_OK);
    EXPECT(mri3d_maxpool3d_fwd(nullptr, P, P, nullptr, nullptr) != MRI3D_OK);
    EXPECT(mri3d_maxpool3d_bwd(nullptr, P, nullptr, P, nullptr) != MRI3D_OK);
    EXPECT(mri3d_upsample3d_fwd(nullptr, P, P, nullptr) != MRI3D_OK);
    EXPECT(mri3d_softmax_dice_fwd(nullptr, P, P, nullptr, nullptr, P, 0, nullptr) != MRI3D_OK);
    for (int c : {1, 8, 16, 48, 96, 128}) {
        Mri3dNormGeom ng;
        memset(&ng, 0, sizeof(ng));
        ng.n = 2; ng.vox = (int64_t)160 * 192 * 160; ng.c = c; ng.x_ld = c; ng.y_ld = c; ng.eps = 1e-5f; ng.group_c = 0;
        for (int inst = 0; inst < 2; ++inst)
            for (int dt = 0; dt < 2; ++dt) {
                ng.instance = inst; ng.dtype = dt;
                EXPECT(mri3d_norm_workspace_bytes(&ng) > 0);
            }
        Mri3dUpGeom ug;
        memset(&ug, 0, sizeof(ug));
        ug.n = 2; ug.di = 40; ug.hi = 48; ug.wi = 40; ug.dout = 80; ug.ho = 96; ug.wo = 80; ug.c = c; ug.x_ld = c; ug.y_ld = c;
        ug.mode = MRI3D_UP_TRILINEAR; ug.rd = ug.rh = ug.rw = 0.5f;
        (void)mri3d_upsample3d_workspace_bytes(&ug);
        Mri3dDiceGeom dg;
        memset(&dg, 0, sizeof(dg));
        dg.n = 2; dg.vox = (int64_t)160 * 192 * 160; dg.c = 2; dg.ct = 1; dg.x_ld = 2; dg.t_ld = 1; dg.eps = 1e-9f;
        EXPECT(mri3d_softmax_dice_workspace_bytes(&dg) > 0);
    }
    EXPECT(mri3d_norm_workspace_bytes(nullptr) == 0);
    EXPECT(mri3d_mask_overlap_workspace_bytes() > 0 && mri3d_order_stats_workspace_bytes() > 0 && mri3d_znorm_workspace_bytes() > 0);
    EXPECT(mri3d_surface_distance_workspace_bytes(160, 192, 160) > 0);
    EXPECT(mri3d_argmax_u8(nullptr, nullptr, 10, 2, 2, 0, nullptr) != MRI3D_OK);
    EXPECT(mri3d_mask_overlap(nullptr, nullptr, 10, nullptr, nullptr, 0, nullptr) != MRI3D_OK);
    EXPECT(mri3d_copy_channels(nullptr, nullptr, 10, 4, 4, 4, 0, nullptr) != MRI3D_OK);
    EXPECT(mri3d_adam_step(nullptr, nullptr, nullptr, nullptr, 10, 1e-3f, 0.9f, 0.999f, 1e-8f, 0.01f, 1, 1.0f, 1, nullptr) != MRI3D_OK);
    {   // patch windows are validated on the host: out-of-volume origins must be refused, long lists must not overrun anything
        std::vector<int32_t> loc(3 * 200, 0);
        loc[3 * 150 + 1] = 1000;   // h origin outside the volume
        EXPECT(mri3d_extract_patches(P, 4, 1, 64, 64, 64, loc.data(), 200, 32, 32, 32, P, nullptr) != MRI3D_OK);
    }
    {   // fused autoencoder operators (round 3): predicates, workspace sizes and every refusal are host-side
        Mri3dConvGeom up;
        memset(&up, 0, sizeof(up));
        up.n = 4; up.di = 160; up.hi = 192; up.wi = 160; up.ci = 8; up.dout = 160; up.ho = 192; up.wo = 160; up.co = 1;
        up.kd = 3; up.kh = 1; up.kw = 1; up.sd = up.sh = up.sw = 1; up.pd = 1; up.dd = up.dh = up.dw = 1; up.x_ld = 8; up.y_ld = 1;
        EXPECT(mri3d_upconv3d_supported(&up, 4) == 1 && mri3d_upconv3d_supported(&up, 2) == 1 && mri3d_upconv3d_supported(&up, 3) == 0);
        EXPECT(mri3d_upconv3d_supported(nullptr, 4) == 0 && mri3d_upconv3d_workspace_bytes(nullptr, 4) == 0);
        EXPECT(mri3d_upconv3d_workspace_bytes(&up, 4) > 0);
        EXPECT(mri3d_upconv3d_fwd(&up, 4, nullptr, (const float*)P, nullptr, P, nullptr) == MRI3D_EINVAL);
        EXPECT(mri3d_upconv3d_fwd(&up, 3, P, (const float*)P, nullptr, P, nullptr) == MRI3D_ENOTSUP);
        EXPECT(mri3d_upconv3d_dgrad(&up, 4, P, nullptr, P, nullptr) == MRI3D_EINVAL);
        std::vector<float> b3(64);
        float* q = b3.data();
        EXPECT(mri3d_upconv3d_fwd(&up, 4, q + 1, q, nullptr, q, nullptr) == MRI3D_EINVAL);                 // x off four channels
        EXPECT(mri3d_upconv3d_wgrad(&up, 4, q, q, q, q, q, 16, nullptr) == MRI3D_EINVAL);                   // workspace too small
        Mri3dConvGeom bad2 = up;
        bad2.sd = 2;
        EXPECT(mri3d_upconv3d_supported(&bad2, 4) == 0);                                                    // strided convolution
        bad2 = up; bad2.ci = 16; bad2.co = 8; bad2.x_ld = 16; bad2.y_ld = 8;
        EXPECT(mri3d_upconv3d_supported(&bad2, 4) == 0);                                                    // no (16, 8, 3) instance
        Mri3dConvGeom first, second;
        memset(&first, 0, sizeof(first));
        first.n = 4; first.di = 160; first.hi = 192; first.wi = 160; first.ci = 1; first.dout = 80; first.ho = 192; first.wo = 160; first.co = 8;
        first.kd = 6; first.kh = 1; first.kw = 1; first.sd = 2; first.sh = first.sw = 1; first.pd = 2; first.dd = first.dh = first.dw = 1;
        first.x_ld = 1; first.y_ld = 8;
        second = first;
        second.di = 80; second.ci = 8; second.dout = 80; second.ho = 96; second.kd = 1; second.kh = 6; second.sd = 1; second.sh = 2;
        second.pd = 0; second.ph = 2; second.x_ld = 8; second.y_ld = 8;
        EXPECT(mri3d_convpair_supported(&first, &second) == 1 && mri3d_convpair_supported(&second, &first) == 0);
        EXPECT(mri3d_convpair_supported(nullptr, &second) == 0 && mri3d_convpair_workspace_bytes(&first, nullptr) == 0);
        EXPECT(mri3d_convpair_workspace_bytes(&first, &second) > 0);
        EXPECT(mri3d_convpair_wgrad_first(&first, &second, nullptr, P, (const float*)P, (float*)P, nullptr, P, 0, nullptr) == MRI3D_EINVAL);
        EXPECT(mri3d_convpair_wgrad_first(&second, &first, P, P, (const float*)P, (float*)P, nullptr, P, 0, nullptr) == MRI3D_ENOTSUP);
        EXPECT(mri3d_convpair_wgrad_first(&first, &second, q, q, q, q, nullptr, q, 16, nullptr) == MRI3D_EINVAL);   // workspace too small
        Mri3dConvGeom s3 = second;
        s3.sh = 1; s3.ho = 191;   // six taps at stride 1 reach a row through six taps: more than the kernel's three slots
        EXPECT(mri3d_convpair_supported(&first, &s3) == 0);
    }
    if (g_fail) {
        fprintf(stderr, "%d host checks failed\n", g_fail);
        return 1;
    }
    printf("host sanitizer driver: %zu workspace queries and the validation branches ran clean\n", checked);
    return 0;
}
