"""ctypes binding of libmri3d_hip.so (C ABI declared in include/mri3d.h).

The library is mandatory for every device tensor: there is no PyTorch/eager fallback.  `lib()` raises if the
shared object is missing instead of silently degrading.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int32, c_int64, c_size_t, c_uint8, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmri3d_hip.so")

F32, BF16 = 0, 1
ACT_NONE, ACT_RELU, ACT_LEAKY, ACT_PRELU = 0, 1, 2, 3
UP_NEAREST, UP_TRILINEAR = 0, 1
PASS_FWD, PASS_DGRAD, PASS_WGRAD = 0, 1, 2


class ConvGeom(Structure):
    _fields_ = [(n, c_int32) for n in (
        "n", "di", "hi", "wi", "ci", "dout", "ho", "wo", "co", "kd", "kh", "kw", "sd", "sh", "sw",
        "pd", "ph", "pw", "dd", "dh", "dw", "x_ld", "y_ld", "dtype")]


class NormGeom(Structure):
    _fields_ = [("n", c_int32), ("vox", c_int64), ("c", c_int32), ("x_ld", c_int32), ("y_ld", c_int32),
                ("instance", c_int32), ("act", c_int32), ("alpha_n", c_int32), ("slope", c_float),
                ("eps", c_float), ("group_c", c_int32), ("dtype", c_int32)]


class PoolGeom(Structure):
    _fields_ = [(n, c_int32) for n in (
        "n", "di", "hi", "wi", "dout", "ho", "wo", "c", "kd", "kh", "kw", "sd", "sh", "sw", "pd", "ph", "pw",
        "x_ld", "y_ld", "dtype")]


class UpGeom(Structure):
    _fields_ = [(n, c_int32) for n in ("n", "di", "hi", "wi", "dout", "ho", "wo", "c", "x_ld", "y_ld", "mode",
                                       "align_corners")] + \
               [("rd", c_float), ("rh", c_float), ("rw", c_float), ("dtype", c_int32)]


class DiceGeom(Structure):
    _fields_ = [("n", c_int32), ("vox", c_int64), ("c", c_int32), ("ct", c_int32), ("x_ld", c_int32),
                ("t_ld", c_int32), ("eps", c_float), ("dtype", c_int32)]


_P = c_void_p
_FP = c_void_p  # float* passed as raw address

# name -> (restype, argtypes); must list every symbol include/mri3d.h declares (tests check this)
SIGNATURES = {
    "mri3d_version": (c_int32, []),
    "mri3d_last_error": (c_char_p, []),
    "mri3d_conv3d_workspace_bytes": (c_size_t, [POINTER(ConvGeom), c_int32]),
    "mri3d_conv3d_fwd": (c_int32, [POINTER(ConvGeom), _P, _P, _P, _P, _P, c_size_t, _P]),
    "mri3d_conv3d_fwd_stats_blocks": (c_int32, [POINTER(ConvGeom)]),
    "mri3d_conv3d_fwd_stats": (c_int32, [POINTER(ConvGeom), _P, _P, _P, _P, _P, _P, c_size_t, _P]),
    "mri3d_conv3d_dgrad": (c_int32, [POINTER(ConvGeom), _P, _P, _P, _P, _P, c_size_t, _P]),
    "mri3d_conv3d_wgrad": (c_int32, [POINTER(ConvGeom), _P, _P, _P, _P, _P, c_size_t, _P]),
    "mri3d_conv3d_cat_supported": (c_int32, [POINTER(ConvGeom), c_int32, c_int32, c_int32]),
    "mri3d_conv3d_fwd_cat": (c_int32, [POINTER(ConvGeom), _P, _P, c_int32, c_int32, _P, _P, _P, _P, _P, c_size_t, _P]),
    "mri3d_conv3d_dgrad_cat": (c_int32, [POINTER(ConvGeom), _P, _P, _P, _P, c_int32, c_int32, _P, c_size_t, _P]),
    "mri3d_conv3d_wgrad_cat": (c_int32, [POINTER(ConvGeom), _P, _P, c_int32, c_int32, _P, _P, _P, _P, c_size_t, _P]),
    "mri3d_conv3d_march_supported": (c_int32, [POINTER(ConvGeom), c_int32]),
    "mri3d_conv3d_march_stats_blocks": (c_int32, [POINTER(ConvGeom)]),
    "mri3d_conv3d_fwd_march": (c_int32, [POINTER(ConvGeom), _P, _P, c_int32, c_int32, _P, _P, _P, _P, _P, c_size_t, _P]),
    "mri3d_conv3d_dgrad_march": (c_int32, [POINTER(ConvGeom), _P, _P, _P, _P, c_int32, c_int32, _P, c_size_t, _P]),
    "mri3d_upconv3d_supported": (c_int32, [POINTER(ConvGeom), c_int32]),
    "mri3d_upconv3d_workspace_bytes": (c_size_t, [POINTER(ConvGeom), c_int32]),
    "mri3d_upconv3d_fwd": (c_int32, [POINTER(ConvGeom), c_int32, _P, _FP, _FP, _P, _P]),
    "mri3d_upconv3d_dgrad": (c_int32, [POINTER(ConvGeom), c_int32, _P, _FP, _P, _P]),
    "mri3d_upconv3d_wgrad": (c_int32, [POINTER(ConvGeom), c_int32, _P, _P, _FP, _FP, _P, c_size_t, _P]),
    "mri3d_convpair_supported": (c_int32, [POINTER(ConvGeom), POINTER(ConvGeom)]),
    "mri3d_convpair_workspace_bytes": (c_size_t, [POINTER(ConvGeom), POINTER(ConvGeom)]),
    "mri3d_convpair_wgrad_first": (c_int32, [POINTER(ConvGeom), POINTER(ConvGeom), _P, _P, _FP, _FP, _FP, _P, c_size_t, _P]),
    "mri3d_norm_workspace_bytes": (c_size_t, [POINTER(NormGeom)]),
    "mri3d_norm_stats": (c_int32, [POINTER(NormGeom), _P, _FP, _FP, _FP, _FP, c_float, _P, c_size_t, _P]),
    "mri3d_norm_stats_from_partials": (c_int32, [POINTER(NormGeom), _P, c_int32, _FP, _FP, _FP, _FP, _FP, c_float, _P]),
    "mri3d_norm_act_fwd": (c_int32, [POINTER(NormGeom), _P, _FP, _FP, _FP, _FP, _FP, _P, _P]),
    "mri3d_norm_act_bwd": (c_int32, [POINTER(NormGeom), c_int32, _P, _P, _FP, _FP, _FP, _FP, _FP, _P, _FP, _FP, _FP,
                                     _P, c_size_t, _P]),
    "mri3d_maxpool3d_fwd": (c_int32, [POINTER(PoolGeom), _P, _P, _P, _P]),
    "mri3d_maxpool3d_bwd": (c_int32, [POINTER(PoolGeom), _P, _P, _P, _P]),
    "mri3d_maxpool3d_bwd_add": (c_int32, [POINTER(PoolGeom), _P, _P, _P, c_int32, _P, _P]),
    "mri3d_upsample3d_workspace_bytes": (c_size_t, [POINTER(UpGeom)]),
    "mri3d_upsample3d_fwd": (c_int32, [POINTER(UpGeom), _P, _P, _P]),
    "mri3d_upsample3d_bwd": (c_int32, [POINTER(UpGeom), _P, _P, _P, c_size_t, _P]),
    "mri3d_softmax_dice_workspace_bytes": (c_size_t, [POINTER(DiceGeom)]),
    "mri3d_softmax_dice_fwd": (c_int32, [POINTER(DiceGeom), _P, _P, _FP, _FP, _P, c_size_t, _P]),
    "mri3d_softmax_dice_bwd": (c_int32, [POINTER(DiceGeom), _P, _P, _FP, _FP, _P, _P]),
    "mri3d_argmax_u8": (c_int32, [_P, _P, c_int64, c_int32, c_int32, c_int32, _P]),
    "mri3d_mask_overlap_workspace_bytes": (c_size_t, []),
    "mri3d_mask_overlap": (c_int32, [_P, _P, c_int64, _P, _P, c_size_t, _P]),
    "mri3d_order_stats_workspace_bytes": (c_size_t, []),
    "mri3d_order_stats_f32": (c_int32, [_P, c_int64, _P, c_int32, _P, _P, c_size_t, _P]),
    "mri3d_piecewise_linear_f32": (c_int32, [_P, _P, c_int64, _P, _P, _P, c_int32, _P]),
    "mri3d_znorm_workspace_bytes": (c_size_t, []),
    "mri3d_znorm_mean_mask_f32": (c_int32, [_P, _P, c_int64, _P, _P, c_size_t, _P]),
    "mri3d_crop_or_pad_f32": (c_int32, [_P, _P, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_float, _P]),
    "mri3d_surface_distance_workspace_bytes": (c_size_t, [c_int32, c_int32, c_int32]),
    "mri3d_surface_distance": (c_int32, [_P, _P, c_int32, c_int32, c_int32, _P, _P, _P, c_size_t, _P]),
    "mri3d_surface_elements": (c_int32, [_P, _P, c_int32, c_int32, c_int32, _P, _P, _P, _P, c_int64, _P, _P, c_size_t, _P]),
    "mri3d_extract_patches": (c_int32, [_P, c_int32, c_int32, c_int32, c_int32, c_int32, _P, c_int32, c_int32, c_int32,
                                        c_int32, _P, _P]),
    "mri3d_aggregate_patches_u8": (c_int32, [_P, _P, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, _P,
                                             c_int32, c_int32, c_int32, c_int32, _P]),
    "mri3d_aggregate_patches_argmax": (c_int32, [_P, c_int32, c_int32, c_int32, _P, c_int32, c_int32, c_int32, c_int32,
                                                 c_int32, c_int32, c_int32, _P, c_int32, c_int32, c_int32, c_int32, _P]),
    "mri3d_copy_channels": (c_int32, [_P, _P, c_int64, c_int32, c_int32, c_int32, c_int32, _P]),
    "mri3d_add_channels": (c_int32, [_P, _P, _P, c_int64, c_int32, c_int32, c_int32, c_int32, c_int32, _P]),
    "mri3d_convert_channels": (c_int32, [_P, c_int32, _P, c_int32, c_int64, c_int32, c_int32, c_int32, _P]),
    "mri3d_adam_step": (c_int32, [_FP, _FP, _FP, _FP, c_int64, c_float, c_float, c_float, c_float, c_float, c_int32,
                                  c_float, c_int32, _P]),
}

_lib = None


class Mri3dError(RuntimeError):
    pass


def lib():
    """Load (once) and return the ctypes handle.  Raises if the .so was not built: no fallback exists."""
    global _lib
    if _lib is None:
        # torch bundles its own libamdhip64.so.7; importing it first makes this library bind to the SAME HIP runtime
        # instance (one soname, one copy per process) so that device pointers and streams are interchangeable.
        import torch  # noqa: F401
        if not os.path.exists(LIB_PATH):
            raise Mri3dError(
                "libmri3d_hip.so not found at %s — run `python -m mri_epilepsy_diagnosis_amd.build` "
                "(or __graft_entry__.build()).  There is no PyTorch fallback for device tensors." % LIB_PATH)
        h = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(h, name)
            fn.restype = res
            fn.argtypes = args
        _lib = h
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().mri3d_last_error()
        raise Mri3dError("%s failed (rc=%d): %s" % (what, rc, msg.decode() if msg else "?"))
