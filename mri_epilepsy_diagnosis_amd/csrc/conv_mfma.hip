// conv_mfma.hip — placeholder until the implicit-GEMM path lands (next commit).
#include "common.h"
namespace mri3d {
bool conv_mfma_supported(const Mri3dConvGeom&, int) { return false; }
size_t conv_mfma_workspace_bytes(const Mri3dConvGeom&, int) { return 0; }
int conv_mfma_fwd(const Mri3dConvGeom&, const float*, const float*, const float*, float*, void*, size_t, hipStream_t) { return MRI3D_ENOTSUP; }
int conv_mfma_dgrad(const Mri3dConvGeom&, const float*, const float*, const float*, float*, void*, size_t, hipStream_t) { return MRI3D_ENOTSUP; }
int conv_mfma_wgrad(const Mri3dConvGeom&, const float*, const float*, float*, float*, void*, size_t, hipStream_t) { return MRI3D_ENOTSUP; }
}
