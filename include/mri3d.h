/*
 * mri3d.h — C ABI of libmri3d_hip.so: the MI355X (gfx950) volumetric operator set behind the
 * 3-D U-Net / separable-conv encoder training path of kondratevakate/mri-epilepsy-diagnosis.
 *
 * The reference has no native layer: its hot path is `model(inputs)` / `loss.backward()` on torch.nn
 * modules (segmentation/routine.py:255-278, classification/routine.py:28-34).  Each entry point below
 * replaces one torch.nn operator that those calls reach; the reference call site is cited per function.
 *
 * Conventions
 *   - All tensors are device pointers into memory owned by the caller (PyTorch's caching allocator).
 *     The library never allocates, frees or synchronises; every kernel is enqueued on `stream`.
 *   - Activations are NDHWC ("channels-last-3d"): element (n,d,h,w,c) of a tensor with pitch `ld`
 *     lives at ((((n*D + d)*H + h)*W + w) * ld + c).  ld >= C lets an operator read or write a channel
 *     slice of a wider buffer (decoder concat buffers) without a copy.
 *   - Weights keep torch's parameter layout (Cout, Cin, kd, kh, kw) so state_dicts stay compatible;
 *     kernels repack into `workspace` on the fly.
 *   - dtype: storage type of the ACTIVATION tensors (x, y, dy, dx, logits ...): MRI3D_F32, or MRI3D_BF16 (bfloat16
 *     storage, fp32 arithmetic and accumulation).  Parameters, statistics, parameter gradients, losses are always fp32.
 *   - Return value: 0 = OK, negative = error; mri3d_last_error() gives a thread-local message.
 *   - Reductions are deterministic (no floating-point atomics).
 */
#ifndef MRI3D_H
#define MRI3D_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* mri3d_stream_t; /* hipStream_t */

enum { MRI3D_OK = 0, MRI3D_EINVAL = -1, MRI3D_ENOTSUP = -2, MRI3D_ELAUNCH = -3, MRI3D_EWORKSPACE = -4 };
enum { MRI3D_F32 = 0, MRI3D_BF16 = 1 };
enum { MRI3D_ACT_NONE = 0, MRI3D_ACT_RELU = 1, MRI3D_ACT_LEAKY = 2, MRI3D_ACT_PRELU = 3 };
enum { MRI3D_UP_NEAREST = 0, MRI3D_UP_TRILINEAR = 1 };
enum { MRI3D_PASS_FWD = 0, MRI3D_PASS_DGRAD = 1, MRI3D_PASS_WGRAD = 2 };

int mri3d_version(void);
const char* mri3d_last_error(void);

/* ------------------------------------------------------------------------------------------------
 * Conv3d — replaces torch.nn.Conv3d as used by unet.UNet (segmentation/routine.py:346-356),
 * AE_model.DownBlock/UpBlock (classification/models/AE_model.py:9-26,74-91), cnn_model (cnn_model.py:14,49-148),
 * Modified3DUNet (segmentation/models/modified_3dunet.py:17-80).
 * ---------------------------------------------------------------------------------------------- */
typedef struct Mri3dConvGeom {
    int32_t n;                 /* batch */
    int32_t di, hi, wi, ci;    /* input  D,H,W,C */
    int32_t dout, ho, wo, co;  /* output D,H,W,C */
    int32_t kd, kh, kw;        /* kernel */
    int32_t sd, sh, sw;        /* stride */
    int32_t pd, ph, pw;        /* zero padding */
    int32_t dd, dh, dw;        /* dilation */
    int32_t x_ld, y_ld;        /* voxel pitch (elements) of the input / output buffers */
    int32_t dtype;             /* MRI3D_F32 */
} Mri3dConvGeom;

size_t mri3d_conv3d_workspace_bytes(const Mri3dConvGeom* g, int pass);

/* y = conv(x, w) + bias.  bias may be NULL. */
int mri3d_conv3d_fwd(const Mri3dConvGeom* g, const void* x, const void* w, const void* bias, void* y,
                     void* workspace, size_t ws_bytes, mri3d_stream_t stream);
/* dx = conv_transpose(dy, w) (+ bias, used when this is the forward of a ConvTranspose3d). dx has pitch x_ld. */
/* Forward convolution that ALSO accumulates the BatchNorm batch statistics of its output (torch.nn.BatchNorm3d in training
 * mode right after torch.nn.Conv3d: `unet.UNet`'s ConvolutionalBlock; segmentation/routine.py:258 -> unet forward): per output
 * channel sum(a) and sum(a^2) of the result a WITHOUT its bias over all N x D x H x W voxels, as float64 partials
 * stat_partials[blocks][co][2] (caller-owned, blocks = mri3d_conv3d_fwd_stats_blocks(g); 0 = geometry not supported, use
 * mri3d_conv3d_fwd + mri3d_norm_stats).  mri3d_norm_stats_from_partials turns them into mean / invstd (shift = bias). */
int32_t mri3d_conv3d_fwd_stats_blocks(const Mri3dConvGeom* g);
int mri3d_conv3d_fwd_stats(const Mri3dConvGeom* g, const void* x, const void* w, const void* bias, void* y,
                           double* stat_partials, void* workspace, size_t ws_bytes, mri3d_stream_t stream);
int mri3d_conv3d_dgrad(const Mri3dConvGeom* g, const void* dy, const void* w, const void* bias, void* dx,
                       void* workspace, size_t ws_bytes, mri3d_stream_t stream);
/* dw (torch layout) = sum_v x (*) dy ; dbias = sum_v dy (dbias may be NULL). */
int mri3d_conv3d_wgrad(const Mri3dConvGeom* g, const void* x, const void* dy, void* dw, void* dbias,
                       void* workspace, size_t ws_bytes, mri3d_stream_t stream);

/* Convolution over torch.cat((x, x2), dim=1) WITHOUT the concatenation — `unet.UNet`'s decoder, x = cat((skip, upsampled))
 * in front of its first ConvolutionalBlock (segmentation/routine.py:346-356 -> unet DecodingBlock.forward).  g describes the
 * concatenated convolution (g->ci = all input channels, g->x_ld = voxel pitch of x); channels [0, split) are read from x,
 * channels [split, ci) from x2 (voxel pitch x2_ld).  The data gradient is written as two dense tensors dx (pitch g->x_ld) and
 * dx2 (pitch dx2_ld); the weight gradient reads both.  Workspace sizes are those of the plain passes
 * (mri3d_conv3d_workspace_bytes).  stat_partials (may be NULL): as in mri3d_conv3d_fwd_stats.
 * mri3d_conv3d_cat_supported(g, split, second_ld, pass) = 1 when the pass is served (3x3x3 / stride 1 / pad 1 on the tiled MFMA
 * kernels, split a multiple of 16, every pointer 16-byte aligned); otherwise concatenate (mri3d_copy_channels) and call the plain
 * entry points. */
int32_t mri3d_conv3d_cat_supported(const Mri3dConvGeom* g, int32_t split, int32_t second_ld, int32_t pass);
int mri3d_conv3d_fwd_cat(const Mri3dConvGeom* g, const void* x, const void* x2, int32_t split, int32_t x2_ld, const void* w,
                         const void* bias, void* y, double* stat_partials, void* workspace, size_t ws_bytes,
                         mri3d_stream_t stream);
int mri3d_conv3d_dgrad_cat(const Mri3dConvGeom* g, const void* dy, const void* w, void* dx, void* dx2, int32_t split,
                           int32_t dx2_ld, void* workspace, size_t ws_bytes, mri3d_stream_t stream);
int mri3d_conv3d_wgrad_cat(const Mri3dConvGeom* g, const void* x, const void* x2, int32_t split, int32_t x2_ld, const void* dy,
                           void* dw, void* dbias, void* workspace, size_t ws_bytes, mri3d_stream_t stream);

/* The d-marching forward / data-gradient kernel BY NAME (csrc/conv_march.hip; nn.Conv3d 3x3x3 / stride 1 / pad 1 of `unet.UNet`,
 * segmentation/routine.py:346-356).  mri3d_conv3d_fwd / _dgrad / _fwd_stats / _fwd_cat / _dgrad_cat choose it themselves for the
 * layers it is faster on; these entry points take every geometry it can compute (mri3d_conv3d_march_supported(g, pass) = 1), so a
 * caller — the parity tests — reaches it with any volume.  x2 / dx2 may be NULL (one tensor; split, *_ld ignored); stat_partials may
 * be NULL, else [mri3d_conv3d_march_stats_blocks(g)][co][2].  Workspace: mri3d_conv3d_workspace_bytes of the pass. */
int32_t mri3d_conv3d_march_supported(const Mri3dConvGeom* g, int32_t pass);
int32_t mri3d_conv3d_march_stats_blocks(const Mri3dConvGeom* g);
int mri3d_conv3d_fwd_march(const Mri3dConvGeom* g, const void* x, const void* x2, int32_t split, int32_t x2_ld, const void* w,
                           const void* bias, void* y, double* stat_partials, void* workspace, size_t ws_bytes,
                           mri3d_stream_t stream);
int mri3d_conv3d_dgrad_march(const Mri3dConvGeom* g, const void* dy, const void* w, void* dx, void* dx2, int32_t split,
                             int32_t dx2_ld, void* workspace, size_t ws_bytes, mri3d_stream_t stream);

/* Conv3d over a nearest-neighbour upsampled input that is never formed (csrc/upconv.hip):
 *     y = conv3d(upsample_nearest(x, scale_factor = scale), w, bias, stride 1, padding g->p*)
 * replaces nn.Upsample(scale_factor=4, mode='nearest') followed by the block's first nn.Conv3d in UpBlock
 * (classification/models/AE_model.py:110-120 with up='upsample'; the last block: Conv3d(8, 1, (3,1,1)) at 160x192x160).
 * g describes the convolution on the VIRTUAL fine input: di, hi, wi = scale x the extents of the coarse tensor x
 * (n, di/scale, hi/scale, wi/scale, ci) whose voxel pitch is x_ld; y / dy have pitch y_ld.  dgrad writes the gradient of the
 * COARSE tensor (each coarse voxel = the sum over its scale^3 fine voxels, fixed summation order).  Served geometries:
 * mri3d_upconv3d_supported(g, scale) = 1 (scale 2 or 4, stride 1, dilation 1, <= 8 taps, the (ci, co, taps) instances listed in
 * upconv.hip); everything else returns MRI3D_ENOTSUP and the caller keeps the two separate operators.  wgrad workspace:
 * mri3d_upconv3d_workspace_bytes. */
int32_t mri3d_upconv3d_supported(const Mri3dConvGeom* g, int32_t scale);
size_t mri3d_upconv3d_workspace_bytes(const Mri3dConvGeom* g, int32_t scale);
int mri3d_upconv3d_fwd(const Mri3dConvGeom* g, int32_t scale, const void* x, const float* w, const float* bias, void* y,
                       mri3d_stream_t stream);
int mri3d_upconv3d_dgrad(const Mri3dConvGeom* g, int32_t scale, const void* dy, const float* w, void* dx, mri3d_stream_t stream);
int mri3d_upconv3d_wgrad(const Mri3dConvGeom* g, int32_t scale, const void* x, const void* dy, float* dw, float* dbias,
                         void* workspace, size_t ws_bytes, mri3d_stream_t stream);

/* The first two convolutions of the autoencoder's first DownBlock (classification/models/AE_model.py:45-53, shipped kwargs conv_k=6,
 * conv_s=2, conv_pad=2): first = Conv3d(1, 8, (6,1,1), stride (s,1,1), padding (p,0,0)) on the input volume, second =
 * Conv3d(8, 8, (1,k,1), stride (1,s,1), padding (0,p,0)) on its output (csrc/sepconv.hip).  mri3d_convpair_wgrad_first computes the
 * FIRST convolution's weight / bias gradient from the gradient dy2 of the SECOND convolution's output (pitch second->y_ld), the
 * second convolution's weight w2 (torch layout) and the input x (pitch first->x_ld): what mri3d_conv3d_dgrad(second) followed by
 * mri3d_conv3d_wgrad(first) compute, without the intermediate gradient tensor.  dw1: (8, 1, 6, 1, 1); dbias1: (8) or NULL.
 * Served pairs: mri3d_convpair_supported = 1 (second's input must be first's output); else MRI3D_ENOTSUP. */
int32_t mri3d_convpair_supported(const Mri3dConvGeom* first, const Mri3dConvGeom* second);
size_t mri3d_convpair_workspace_bytes(const Mri3dConvGeom* first, const Mri3dConvGeom* second);
int mri3d_convpair_wgrad_first(const Mri3dConvGeom* first, const Mri3dConvGeom* second, const void* x, const void* dy2, const float* w2,
                               float* dw1, float* dbias1, void* workspace, size_t ws_bytes, mri3d_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * BatchNorm3d / InstanceNorm3d fused with the following activation — replaces
 * nn.BatchNorm3d + nn.PReLU (unet.UNet ConvolutionalBlock), nn.BatchNorm3d + LeakyReLU/ReLU
 * (AE_model.py:30-36, cnn_model.py), nn.InstanceNorm3d + LeakyReLU (modified_3dunet.py:20-94).
 * groups = 1 (batch norm: statistics over n*vox per channel) or n (instance norm: per (n,c)); GroupNorm
 * (nn.GroupNorm(4, C) of segmentation/models/unet3d.py:12) = instance mode with group_c = C/4 channels pooled.
 * ---------------------------------------------------------------------------------------------- */
typedef struct Mri3dNormGeom {
    int32_t n;        /* batch */
    int64_t vox;      /* D*H*W */
    int32_t c;        /* channels */
    int32_t x_ld;     /* voxel pitch of x / dx */
    int32_t y_ld;     /* voxel pitch of y / dy */
    int32_t instance; /* 0 = batch statistics, 1 = per-instance statistics */
    int32_t act;      /* MRI3D_ACT_* applied after the affine transform */
    int32_t alpha_n;  /* PReLU: number of alpha parameters (1 or c) */
    float slope;      /* LeakyReLU negative slope */
    float eps;
    int32_t group_c;  /* GroupNorm: channels per group (needs instance = 1; statistics per (n, group)); 0 otherwise */
    int32_t dtype;
} Mri3dNormGeom;

size_t mri3d_norm_workspace_bytes(const Mri3dNormGeom* g);

/* mean/invstd: [groups*c] floats where groups = instance ? n : 1.
 * If running_mean/running_var are non-NULL (batch mode only) they are updated in place:
 *   running = (1-momentum)*running + momentum*stat, with the unbiased variance, as torch does. */
int mri3d_norm_stats(const Mri3dNormGeom* g, const void* x, float* mean, float* invstd,
                     float* running_mean, float* running_var, float momentum,
                     void* workspace, size_t ws_bytes, mri3d_stream_t stream);
/* mean / invstd (and the running-statistics update of mri3d_norm_stats) from float64 partial sums
 * partials[nblk][c][2] = (sum(x - shift[c]), sum((x - shift[c])^2)) over disjoint parts of the N x vox voxels (batch statistics
 * only).  shift may be NULL (zero). */
int mri3d_norm_stats_from_partials(const Mri3dNormGeom* g, const double* partials, int32_t nblk, const float* shift,
                                   float* mean, float* invstd, float* running_mean, float* running_var, float momentum,
                                   mri3d_stream_t stream);
/* y = act(gamma * (x - mean) * invstd + beta).  gamma/beta may be NULL (affine=False); alpha is the
 * PReLU parameter (device pointer) when act == MRI3D_ACT_PRELU.  mean/invstd NULL => identity norm. */
int mri3d_norm_act_fwd(const Mri3dNormGeom* g, const void* x, const float* mean, const float* invstd,
                       const float* gamma, const float* beta, const float* alpha, void* y,
                       mri3d_stream_t stream);
/* Backward of norm_act_fwd.  training != 0: statistics were computed from x (batch/instance mode);
 * training == 0: mean/invstd are constants (eval-mode BatchNorm).  dgamma/dbeta/dalpha may be NULL. */
int mri3d_norm_act_bwd(const Mri3dNormGeom* g, int training, const void* x, const void* dy,
                       const float* mean, const float* invstd, const float* gamma, const float* beta,
                       const float* alpha, void* dx, float* dgamma, float* dbeta, float* dalpha,
                       void* workspace, size_t ws_bytes, mri3d_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * MaxPool3d — nn.MaxPool3d(2) in unet.UNet / AE_model.py:27 / cnn_model.py:115-148, (4,2) in cnn_model.py:221,232.
 * idx: one byte per output element = window-local offset (kd,kh,kw raster) of the arg-max.
 * ---------------------------------------------------------------------------------------------- */
typedef struct Mri3dPoolGeom {
    int32_t n, di, hi, wi, dout, ho, wo, c;
    int32_t kd, kh, kw, sd, sh, sw, pd, ph, pw;
    int32_t x_ld, y_ld;
    int32_t dtype;
} Mri3dPoolGeom;
int mri3d_maxpool3d_fwd(const Mri3dPoolGeom* g, const void* x, void* y, uint8_t* idx, mri3d_stream_t stream);
int mri3d_maxpool3d_bwd(const Mri3dPoolGeom* g, const void* dy, const uint8_t* idx, void* dx, mri3d_stream_t stream);
/* dx = maxpool3d_bwd(dy) + addend: `addend` (same storage type, voxel pitch addend_ld >= c) is the other gradient of the pool's
 * input when that tensor also feeds a skip connection (unet.UNet encoder: `skip = x; x = pool(x)`), which autograd would
 * otherwise sum in a separate full-resolution pass. */
int mri3d_maxpool3d_bwd_add(const Mri3dPoolGeom* g, const void* dy, const uint8_t* idx, const void* addend,
                            int32_t addend_ld, void* dx, mri3d_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Upsample — nn.Upsample(scale_factor=2, mode='trilinear', align_corners=False) (unet.UNet decoder),
 * mode='nearest' (modified_3dunet.py:13, AE_model.py:70-73), F.interpolate(size=...) (AE_model.py:119).
 * rd/rh/rw: source step per destination step (torch's "scale" = 1/scale_factor, or in/out for size=).
 * ---------------------------------------------------------------------------------------------- */
typedef struct Mri3dUpGeom {
    int32_t n, di, hi, wi, dout, ho, wo, c;
    int32_t x_ld, y_ld;
    int32_t mode;          /* MRI3D_UP_* */
    int32_t align_corners; /* trilinear only */
    float rd, rh, rw;
    int32_t dtype;
} Mri3dUpGeom;
size_t mri3d_upsample3d_workspace_bytes(const Mri3dUpGeom* g);
int mri3d_upsample3d_fwd(const Mri3dUpGeom* g, const void* x, void* y, mri3d_stream_t stream);
int mri3d_upsample3d_bwd(const Mri3dUpGeom* g, const void* dy, void* dx, void* workspace, size_t ws_bytes,
                         mri3d_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Fused softmax(dim=C) + soft-Dice loss — F.softmax + get_dice_loss + .mean()
 * (segmentation/routine.py:239-253,272-274).  target has ct = 1 (broadcast over classes, the
 * reference's behaviour) or ct = c channels.  stats: [n*c*3] floats (tp, sum_p, sum_g) kept for bwd.
 * ---------------------------------------------------------------------------------------------- */
typedef struct Mri3dDiceGeom {
    int32_t n;
    int64_t vox;
    int32_t c, ct;
    int32_t x_ld, t_ld;
    float eps;
    int32_t dtype;
} Mri3dDiceGeom;
size_t mri3d_softmax_dice_workspace_bytes(const Mri3dDiceGeom* g);
int mri3d_softmax_dice_fwd(const Mri3dDiceGeom* g, const void* logits, const void* target, float* loss,
                           float* stats, void* workspace, size_t ws_bytes, mri3d_stream_t stream);
/* dlogits = dloss * d(loss)/d(logits); dloss is a device pointer to one float. */
int mri3d_softmax_dice_bwd(const Mri3dDiceGeom* g, const void* logits, const void* target, const float* stats,
                           const float* dloss, void* dlogits, mri3d_stream_t stream);

/* argmax over channels -> uint8 mask (validate_dsc_asd, segmentation/routine.py:226-227). First max wins. */
int mri3d_argmax_u8(const void* logits, uint8_t* out, int64_t nvox, int32_t c, int32_t ld, int32_t dtype,
                    mri3d_stream_t stream);

/* Overlap counts of two uint8 masks for the validation metrics of validate_dsc_asd (segmentation/routine.py:198-235;
 * compute_dice_coefficient segmentation/metrics.py:312-329; get_iou_score routine.py:198-203), exact integers:
 *   counts[0] = sum(gt)  [1] = sum(pred)  [2] = sum(gt & pred)  [3] = #(gt>0 and pred>0)  [4] = #(gt>0 or pred>0)
 *   Dice = 2*counts[2] / (counts[0] + counts[1])   (NaN when the denominator is 0);   IoU = counts[3] / counts[4]. */
size_t mri3d_mask_overlap_workspace_bytes(void);
int mri3d_mask_overlap(const uint8_t* pred, const uint8_t* gt, int64_t nvox, int64_t* counts, void* workspace,
                       size_t ws_bytes, mri3d_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Intensity standardisation ahead of the path: the reference's collate function runs `normalize`
 * (classification/train_ENC_CLF.ipynb cell 9) on every volume — np.percentile at 13 percentiles over all voxels, then a
 * piecewise-linear map between landmarks evaluated in float64.
 *   order_stats: out[i] = the ranks[i]-th smallest element (0-based) of x[0..n); ranks is a HOST array of nranks <= 32.
 *                Exact (integer radix histograms).  np.percentile's linear interpolation between two neighbouring order
 *                statistics is left to the caller (13 float64 operations).
 *   piecewise_linear: y = (float)(slope[b] * (double)x + intercept[b]), b = #{e : edges[e] <= x} (np.digitize, right=False),
 *                multiply and add rounded separately as numpy does; edges / slope / intercept are HOST arrays,
 *                nseg <= 16 segments, nseg-1 non-decreasing edges.
 * ---------------------------------------------------------------------------------------------- */
size_t mri3d_order_stats_workspace_bytes(void);
int mri3d_order_stats_f32(const float* x, int64_t n, const int64_t* ranks, int32_t nranks, float* out, void* workspace,
                          size_t ws_bytes, mri3d_stream_t stream);
int mri3d_piecewise_linear_f32(const float* x, float* y, int64_t n, const double* edges, const double* slope,
                               const double* intercept, int32_t nseg, mri3d_stream_t stream);
/* TorchIO transforms the notebooks apply after the histogram standardisation (segmentation/pretraining_3d_unet.ipynb cell 9:
 * `ZNormalization(masking_method=ZNormalization.mean)`, `CropOrPad(...)`).  Third-party arithmetic, restated from its
 * documentation ("parity unpinned"):
 *   znorm_mean_mask: mask = x > mean(x);  y = (x - mean(x[mask])) / std(x[mask]) (unbiased);  stats (device, 4 doubles) =
 *                    {mean of all voxels, masked count, masked mean, masked std}.
 *   crop_or_pad:     centred crop / zero-pad of `outer` contiguous (di,hi,wi) volumes to (dout,ho,wo); the odd voxel of an
 *                    uneven difference goes to the far end (ini = floor(diff/2)). */
size_t mri3d_znorm_workspace_bytes(void);
int mri3d_znorm_mean_mask_f32(const float* x, float* y, int64_t n, double* stats, void* workspace, size_t ws_bytes,
                              mri3d_stream_t stream);
int mri3d_crop_or_pad_f32(const float* x, float* y, int32_t outer, int32_t di, int32_t hi, int32_t wi, int32_t dout,
                          int32_t ho, int32_t wo, float fill, mri3d_stream_t stream);

/* Average surface distance inputs of validate_dsc_asd (segmentation/routine.py:205-214): for two (d,h,w) uint8 masks
 * (non-zero = inside) and the 256-entry surface-element area table of metrics.py:57-71 (HOST array, for the spacing in use)
 *   sums[0] = sum(dist_to_pred_surface * area), sums[1] = sum(area) over the surface elements of gt,
 *   sums[2], sums[3] = the same over the surface elements of pred with distances to the gt surface       (device doubles)
 * with surface elements and areas as compute_surface_distances defines them (segmentation/metrics.py:25-178) and the exact
 * Euclidean distance transform for unit spacing.  compute_average_surface_distance = (sums[0]/sums[1], sums[2]/sums[3]). */
size_t mri3d_surface_distance_workspace_bytes(int32_t d, int32_t h, int32_t w);
int mri3d_surface_distance(const uint8_t* gt, const uint8_t* pred, int32_t d, int32_t h, int32_t w,
                           const double* area_table, double* sums, void* workspace, size_t ws_bytes, mri3d_stream_t stream);
/* The surface-element lists behind compute_surface_distances (segmentation/metrics.py:25-178), for the order-dependent metrics
 * (compute_robust_hausdorff, compute_surface_overlap_at_tolerance, compute_surface_dice_at_tolerance, metrics.py:208-310):
 * per direction, the exact squared distance of every surface element to the other surface (>= 0x3f000000: the other mask
 * has no surface) and its neighbour code, in arbitrary order; counts[0], counts[1] (device) = list lengths (entries beyond
 * `capacity` are dropped: size the lists for (d+1)(h+1)(w+1) to be safe).  Same workspace as mri3d_surface_distance. */
int mri3d_surface_elements(const uint8_t* gt, const uint8_t* pred, int32_t d, int32_t h, int32_t w, int32_t* d2_gt,
                           uint8_t* code_gt, int32_t* d2_pred, uint8_t* code_pred, int64_t capacity, uint64_t* counts,
                           void* workspace, size_t ws_bytes, mri3d_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Patch pipeline: the windows TorchIO cuts for patch training and grid inference
 * (torchio.Queue + torchio.sampler.ImageSampler, segmentation/routine.py:150-178 and pretraining_3d_unet.ipynb cell 24;
 * torchio.inference.GridSampler / GridAggregator.add_batch(labels, locations), pretraining_3d_unet.ipynb cell 26).
 * Third-party arithmetic, restated in oracle/patches.py ("parity unpinned").  Volumes are `nvol` contiguous (d,h,w) arrays
 * of `elem_bytes`-sized elements in HBM; `loc_host` is a HOST table of npatch x 4 int32 = (volume index, d0, h0, w0), checked
 * here against the volume (MRI3D_EINVAL when a window leaves it) and handed to the kernels by value.
 *   extract_patches:          out[p, z, y, x] = volumes[loc[p].vol, d0+z, h0+y, w0+x]
 *   aggregate_patches_u8:     each (pd,ph,pw) uint8 label window is cropped by `border` voxels on all six faces and written
 *                             to out[vol] at its location, in patch order: where cropped windows overlap the later patch
 *                             wins (what sequential add_batch slicing does); voxels no cropped window covers are untouched.
 *   aggregate_patches_argmax: the same, reading logits [p][z][y][x][ld] and taking argmax over c channels in flight
 *                             (labels = logits.argmax(dim=1, keepdim=True); first maximum wins).
 * ---------------------------------------------------------------------------------------------- */
int mri3d_extract_patches(const void* volumes, int32_t elem_bytes, int32_t nvol, int32_t d, int32_t h, int32_t w,
                          const int32_t* loc_host, int32_t npatch, int32_t pd, int32_t ph, int32_t pw, void* out,
                          mri3d_stream_t stream);
int mri3d_aggregate_patches_u8(const uint8_t* patches, const int32_t* loc_host, int32_t npatch, int32_t pd, int32_t ph,
                               int32_t pw, int32_t bd, int32_t bh, int32_t bw, uint8_t* out, int32_t nvol, int32_t d,
                               int32_t h, int32_t w, mri3d_stream_t stream);
int mri3d_aggregate_patches_argmax(const void* logits, int32_t c, int32_t ld, int32_t dtype, const int32_t* loc_host,
                                   int32_t npatch, int32_t pd, int32_t ph, int32_t pw, int32_t bd, int32_t bh, int32_t bw,
                                   uint8_t* out, int32_t nvol, int32_t d, int32_t h, int32_t w, mri3d_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Channel-slice plumbing: torch.cat along channels (unet.UNet decoder, modified_3dunet.py:158-178) and
 * residual adds (modified_3dunet.py:108, cnn_model.py:34).
 *   copy: dst[v, 0:c] = src[v, 0:c]      add: dst[v, 0:c] = a[v,0:c] + b[v,0:c]
 * ---------------------------------------------------------------------------------------------- */
int mri3d_copy_channels(const void* src, void* dst, int64_t nvox, int32_t c, int32_t src_ld, int32_t dst_ld,
                        int32_t dtype, mri3d_stream_t stream);
int mri3d_add_channels(const void* a, const void* b, void* dst, int64_t nvox, int32_t c, int32_t a_ld,
                       int32_t b_ld, int32_t dst_ld, int32_t dtype, mri3d_stream_t stream);
/* Storage-type cast at the edge of a bf16 region (the `.to(bfloat16)` / `.float()` an autocast region inserts;
 * BASELINE configs[3]): dst[v, 0:c] = (dst_dtype) src[v, 0:c], round-to-nearest-even towards bf16. */
int mri3d_convert_channels(const void* src, int32_t src_dtype, void* dst, int32_t dst_dtype, int64_t nvox, int32_t c,
                           int32_t src_ld, int32_t dst_ld, mri3d_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * AdamW / Adam on one flat fp32 buffer — torch.optim.AdamW (segmentation/routine.py:358) and
 * torch.optim.Adam with L2 weight_decay (classification/routine.py:271,275).
 * grad_scale multiplies the gradient first (1/world_size after the RCCL sum all-reduce).
 * decoupled != 0: AdamW (p *= 1 - lr*wd); decoupled == 0: Adam (g += wd*p).
 * ---------------------------------------------------------------------------------------------- */
int mri3d_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                    float eps, float weight_decay, int32_t step, float grad_scale, int32_t decoupled,
                    mri3d_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MRI3D_H */
