"""ctypes binding of libdcs_hip.so (the C ABI declared in include/dcs_hip.h).

The product path has NO fallback: if the shared library is missing or a symbol
is absent, importing/using it raises.  Build it with
``make -C doubly-contrastive-semseg_amd/dcs_amd/csrc`` or ``__graft_entry__.build()``.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# DCS_LIB: another build of the same library (compiler experiments, tools/slp_repro.sh); read once, at import
LIB_PATH = os.environ.get("DCS_LIB") or os.path.join(_HERE, "libdcs_hip.so")
MAX_TAPS = 49


class DcsConvGeom(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "N", "SH", "SW", "DH", "DW", "TY", "TX", "sy", "sx", "dsy", "dsx", "dy0", "dx0", "K", "Cout",
        "ntaps", "wstride", "src_cstride", "dst_cstride", "stem")] + [
        ("offy", C.c_int16 * MAX_TAPS), ("offx", C.c_int16 * MAX_TAPS), ("wofs", C.c_int32 * MAX_TAPS)]


MULTI_MAX = 3             # DCS_MULTI_MAX


class DcsGatherLaunch(C.Structure):
    """One sub-launch of dcs_conv_gather_x3_multi / dcs_conv3x3_x3w_multi (include/dcs_hip.h)."""
    _fields_ = [("src", C.c_void_p), ("wgt", C.c_void_p), ("bias", C.c_void_p), ("dst", C.c_void_p),
                ("geom", C.POINTER(DcsConvGeom)), ("stats", C.c_void_p), ("pro", C.c_void_p), ("bn_y", C.c_void_p),
                ("bn_mask", C.c_void_p), ("bn", C.c_void_p), ("slab_stride", C.c_int64), ("accumulate", C.c_int32),
                ("relu", C.c_int32), ("nsplit", C.c_int32), ("src_max", C.c_void_p)]


class DcsWgradLaunch(C.Structure):
    """One sub-launch of dcs_conv_wgrad_x3_multi."""
    _fields_ = [("src", C.c_void_p), ("dy", C.c_void_p), ("slab", C.c_void_p), ("geom", C.POINTER(DcsConvGeom)),
                ("pro", C.c_void_p), ("dy_cstride", C.c_int32), ("split0", C.c_int32), ("nsplit", C.c_int32),
                ("dy_max", C.c_void_p)]


_P = C.c_void_p
_I = C.c_int
_L = C.c_int64
_F = C.c_float
_D = C.c_double
_G = C.POINTER(DcsConvGeom)

# name -> argtypes (all return int); mirrors include/dcs_hip.h one to one
SIGNATURES = {
    "dcs_set_option": [C.c_char_p, _I],
    "dcs_get_option": [C.c_char_p, C.POINTER(C.c_int)],
    "dcs_conv_gather": [_P, _P, _P, _P, _G, _I, _P, _P],
    "dcs_conv_gather_bnbwd": [_P, _P, _P, _G, _I, _P, _P, _P, _I, _P, _P],
    "dcs_conv_gather_split": [_P, _P, _P, C.POINTER(DcsConvGeom), _I, _L, _P],
    "dcs_conv_wgrad": [_P, _P, _P, _G, _I, _I, _I, _P],
    "dcs_split_weight": [_P, _P, _L, _I, _P],
    "dcs_conv_gather_x3": [_P, _P, _P, _P, _G, _I, _P, _P, _P, _P, _P, _I, _I, _L, _P, _P],
    "dcs_split_weight_h2": [_P, _P, _L, _I, _P],
    "dcs_split_weight_frag": [_P, _P, _L, _I, _P],
    "dcs_split_weight_frag_h2": [_P, _P, _L, _I, _P],
    "dcs_conv3x3_x3w": [_P, _P, _P, _P, _G, _I, _P, _P, _P, _P, _P, _I, _P, _P],
    "dcs_conv_wgrad_x3": [_P, _P, _P, _G, _I, _I, _I, _P, _P, _P],
    "dcs_conv_gather_x3_multi": [C.POINTER(DcsGatherLaunch), _I, _P],
    "dcs_conv3x3_x3w_multi": [C.POINTER(DcsGatherLaunch), _I, _P],
    "dcs_conv_wgrad_x3_multi": [C.POINTER(DcsWgradLaunch), _I, _P],
    "dcs_conv_gather_pro": [_P, _P, _P, _P, _G, _I, _P, _P, _I, _L, _P],
    "dcs_conv_wgrad_pro": [_P, _P, _P, _G, _I, _I, _I, _P, _P],
    "dcs_reduce_slab": [_P, _P, _L, _I, _I, _I, _I, _P],
    "dcs_pack_dgrad_weight": [_P, _P, _I, _I, _I, _I, _I, _I, _P],
    "dcs_pack_stem_weight": [_P, _P, _I, _I, _P],
    "dcs_transpose": [_P, _P, _I, _I, _P],
    "dcs_colsum_partial": [_P, _P, _P, _P, _P, _I, _L, _I, _I, _I, _I, _I, _P],
    "dcs_colsum_final": [_P, _P, _I, _I, _I, _F, _D, _P],
    "dcs_bn_finalize": [_P, _P, _P, _P, _P, _P, _I, _D, _F, _F, _I, _I, _P],
    "dcs_bn_ema_again": [_P, _P, _P, _I, _D, _F, _F, _P],
    "dcs_maxabs": [_P, _L, _P, _P],
    "dcs_bn_act": [_P, _P, _P, _P, _P, _L, _I, _I, _P, _P],
    "dcs_bn_bwd_apply": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _I, _I, _I, _I, _P, _P, _P],
    "dcs_normalize_pyramid": [_P, _P, _P, _P, _I, _I, _I, _P, _P, _P],
    "dcs_bn_relu_maxpool": [_P, _P, _P, _P, _I, _I, _I, _I, _P],
    "dcs_maxpool_bwd": [_P, _P, _P, _I, _I, _I, _I, _P],
    "dcs_bn_pool_bwd_partial": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "dcs_bn_pool_bwd_partial_pooled": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "dcs_bn_pool_bwd_apply": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P],
    "dcs_upsample_add": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "dcs_upsample_add_stats": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "dcs_upsample_bwd": [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P],
    "dcs_upsample_to_nchw": [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "dcs_upsample_to_nchw_bwd": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "dcs_seg_loss": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _I, _I, _P],
    "dcs_seg_loss_fused": [_P, _I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _F, _I, _I, _P],
    "dcs_seg_loss_fused_blocks": [_I, _I, _I],
    "dcs_seg_loss_final": [_P, _P, _I, _P],
    "dcs_scale_inplace": [_P, _L, _P, _P, _P],
    "dcs_anchor_keys": [_P, _I, _I, _P, _I, _I, _I, _I, _I, _I, _P, _P, _I, _P],
    "dcs_anchor_select": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "dcs_sampler_plan": [_P, _I, _I, _I, _I, _P, C.POINTER(C.c_int), _P, _P, _P, C.POINTER(C.c_int)],
    "dcs_gather_rows": [_P, _P, _P, _I, _I, _P],
    "dcs_scatter_add_rows": [_P, _P, _P, _I, _I, _P],
    "dcs_gather_rows_bilinear": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "dcs_scatter_rows_bilinear": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "dcs_label_boundary_weights": [_P, _P, _P, _P, _I, _I, _I, _I, _L, _P],
    "dcs_confusion": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "dcs_contrast_fused_ws": [_I, _I, C.POINTER(C.c_int64)],
    "dcs_contrast_fused": [_P, _I, _P, _I, _P, _I, _I, _I, _I, _F, _P, _P, _I, _P, _I, _P, _L, _P, _P],
    "dcs_contrast_rows": [_P, _P, _P, _P, _I, _I, _I, _F, _P],
    "dcs_symmetrize": [_P, _P, _I, _I, _P],
    "dcs_sum_scalar": [_P, _P, _I, _F, _P],
    "dcs_dropout": [_P, _P, _P, _P, _L, _F, C.c_uint32, _P],
    "dcs_dropout_bwd": [_P, _P, _P, _L, _F, _P],
    "dcs_adam_step": [_P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _I, _P],
    "dcs_axpy": [_P, _P, _L, _F, _P],
    "dcs_add_rowvec_bcast": [_P, _P, _I, _L, _I, _F, _I, _P],
    "dcs_relu_bwd_rows": [_P, _P, _P, _L, _P],
}

ERRORS = {-1: "DCS_E_ARG (bad shape / alignment / null pointer)", -2: "DCS_E_LAUNCH (HIP launch failed)",
          -3: "DCS_E_UNSUPPORTED"}

_lib = None


def load():
    """Load libdcs_hip.so and bind every symbol of the C ABI; raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the HIP library is required (no CPU fallback). "
            "Build it with `python -c 'import __graft_entry__ as g; g.build()'` or `make -C .../dcs_amd/csrc`.")
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.argtypes = argtypes
        fn.restype = C.c_int
    lib.dcs_version.restype = C.c_char_p
    lib.dcs_version.argtypes = []
    _lib = lib
    return lib


def check(rc: int, name: str):
    if rc != 0:
        raise RuntimeError(f"{name} failed: {ERRORS.get(rc, rc)}")


def set_option(name: str, value: int) -> int:
    """Change a library switch (include/dcs_hip.h: dcs_set_option); returns the previous value."""
    lib = load()
    old = C.c_int(0)
    check(lib.dcs_get_option(name.encode(), C.byref(old)), "dcs_get_option")
    check(lib.dcs_set_option(name.encode(), int(value)), "dcs_set_option")
    return old.value


def get_option(name: str) -> int:
    v = C.c_int(0)
    check(load().dcs_get_option(name.encode(), C.byref(v)), "dcs_get_option")
    return v.value
