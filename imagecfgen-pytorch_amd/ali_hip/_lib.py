"""ctypes binding of libali_hip.so (C ABI declared in include/ali_hip.h).

The library is the product: if it cannot be loaded every GPU entry point raises
``AliHipUnavailable`` -- there is no CPU or eager-PyTorch fallback for CUDA tensors.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int32, c_int64, c_size_t, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libali_hip.so")


class AliHipUnavailable(RuntimeError):
    pass


class AliConvGeom(Structure):
    _fields_ = [(n, c_int32) for n in ("B", "H", "W", "C", "P", "Q", "K", "R", "S", "stride", "pad")]


class AliEpilogue(Structure):
    _fields_ = [("bias", c_void_p), ("act", c_int32), ("slope", c_float), ("mask", c_void_p), ("mask_ld", c_int32),
                ("dact_y", c_void_p), ("dact", c_int32), ("dslope", c_float),
                # fused BatchNorm reductions (include/ali_hip.h)
                ("bn_part", c_void_p), ("bn_mode", c_int32), ("bn_groups", c_int32), ("bn_stat_mask", c_void_p),
                ("bn_mask_ld", c_int32), ("bn_x", c_void_p), ("bn_mean", c_void_p), ("bn_invstd", c_void_p),
                ("bn_mask_in", c_void_p), ("bn_mask_pre", c_void_p), ("mfma_f16", c_int32),
                ("in16", c_void_p), ("w16", c_void_p), ("out16", c_void_p),
                ("tile_order", c_void_p), ("tile_order_n", c_int32), ("in_ld", c_int32), ("out_ld", c_int32),
                ("in_ch_live", c_int32), ("bn_slots", c_int32), ("dact_y16", c_void_p)]


class AliWgradFold(Structure):
    _fields_ = [("ws", c_void_p), ("dst", c_void_p), ("dbws", c_void_p), ("db", c_void_p),
                ("slab", c_int64), ("s_dc", c_int64), ("s_gc", c_int64), ("s_tap", c_int64),
                ("S", c_int32), ("Mtot", c_int32), ("Cg", c_int32), ("Cd", c_int32), ("Cg_log", c_int32),
                ("Cd_log", c_int32), ("T", c_int32), ("reserved", c_int32), ("ws_used", c_uint64)]


class AliWgradJob(Structure):
    _fields_ = [("opaque", c_uint64 * 40)]


class AliGemmJob(Structure):
    _fields_ = [("opaque", c_uint64 * 112)]


ACT_NONE, ACT_LEAKY, ACT_TANH = 0, 1, 2

# name -> (restype, argtypes); every symbol include/ali_hip.h declares
SIGNATURES = {
    "ali_conv_workspace_bytes": (c_size_t, [POINTER(AliConvGeom), c_int32]),
    "ali_conv_mtiles": (c_int32, [POINTER(AliConvGeom), c_int32, c_int32, POINTER(c_int32), POINTER(c_int32)]),
    "ali_conv_writes_out16": (c_int32, [POINTER(AliConvGeom), c_int32]),
    "ali_conv_uses_f16": (c_int32, [POINTER(AliConvGeom), c_int32, POINTER(AliEpilogue)]),
    "ali_conv_tile_order": (c_int32, [POINTER(AliConvGeom), c_int32, c_int32, c_void_p, c_int32]),
    "ali_conv_fwd": (c_int32, [POINTER(AliConvGeom), c_void_p, c_void_p, c_void_p, POINTER(AliEpilogue), c_void_p,
                               c_size_t, c_void_p]),
    "ali_conv_bwd_data": (c_int32, [POINTER(AliConvGeom), c_void_p, c_void_p, c_void_p, POINTER(AliEpilogue),
                                    c_void_p, c_size_t, c_void_p]),
    "ali_conv_bwd_weight": (c_int32, [POINTER(AliConvGeom), c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int64,
                                      c_int64, c_int64, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_int32,
                                      POINTER(AliWgradFold), POINTER(AliWgradJob), c_int32, c_void_p, c_size_t, c_void_p]),
    "ali_conv_fwd_job": (c_int32, [POINTER(AliConvGeom), c_void_p, c_void_p, c_void_p, POINTER(AliEpilogue), c_void_p,
                                   c_size_t, POINTER(AliGemmJob), c_void_p]),
    "ali_conv_bwd_data_job": (c_int32, [POINTER(AliConvGeom), c_void_p, c_void_p, c_void_p, POINTER(AliEpilogue),
                                        c_void_p, c_size_t, POINTER(AliGemmJob), c_void_p]),
    "ali_gemm_launch_multi": (c_int32, [c_int32, POINTER(AliGemmJob), c_void_p]),
    "ali_wgrad_deferrable": (c_int32, [POINTER(AliConvGeom), c_int32]),
    "ali_wgrad_launch_multi": (c_int32, [c_int32, POINTER(AliWgradJob), c_void_p]),
    "ali_wgrad_fold_multi": (c_int32, [c_int32, POINTER(AliWgradFold), c_void_p]),
    "ali_wgrad_pixtab": (c_int32, [POINTER(AliConvGeom), c_void_p, c_void_p]),
    "ali_pack_weights": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int64, c_int64, c_int64,
                                   c_void_p]),
    "ali_pack_weights_multi": (c_int32, [c_int32, POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p),
                                         POINTER(c_int32), POINTER(c_int64), c_void_p]),
    "ali_act_bwd": (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_float, c_void_p]),
    "ali_colsum": (c_int32, [c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p, c_size_t, c_void_p]),
    "ali_rowmask_mul": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p]),
    "ali_dropout_mask": (c_int32, [c_uint64, c_uint64, c_void_p, c_float, c_void_p, c_int64, c_void_p]),
    "ali_dropout_mask_multi": (c_int32, [c_uint64, c_void_p, POINTER(c_int64), POINTER(c_float), POINTER(c_int32),
                                         POINTER(c_int32), c_int32, c_void_p, c_void_p]),
    "ali_bn_stats": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p,
                               c_float, c_float, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int64,
                               c_void_p, c_size_t, c_void_p]),
    "ali_bn_apply": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32,
                               c_int32, c_int64, c_void_p]),
    "ali_bn_bwd": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32,
                             c_int32, c_int32, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "ali_bn_stats_from_partials": (c_int32, [c_void_p, c_int32, c_int32, c_int32, c_int64, c_void_p, c_void_p, c_void_p,
                                             c_void_p, c_float, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                             c_void_p]),
    "ali_bn_bwd_from_partials": (c_int32, [c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                           c_void_p, c_int32, c_int32, c_int32, c_int32, c_float, c_void_p, c_void_p,
                                           c_void_p, c_void_p]),
    "ali_bce_logits": (c_int32, [c_void_p, c_int32, c_float, c_float, c_void_p, c_void_p, c_void_p]),
    "ali_bce_logits_pair": (c_int32, [c_void_p, c_int32, c_float, c_float, c_float, c_void_p, c_void_p, c_void_p]),
    "ali_attr_pack": (c_int32, [POINTER(c_void_p), POINTER(c_int32), POINTER(c_int32), c_int32, POINTER(c_void_p), c_int32,
                                c_int32, c_void_p, c_void_p, c_void_p]),
    "ali_g_input": (c_int32, [c_void_p, c_int32, POINTER(c_void_p), POINTER(c_int32), POINTER(c_int32), POINTER(c_void_p),
                              c_int32, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p]),
    "ali_g_input_table_grad": (c_int32, [c_void_p, c_int32, c_int32, c_void_p, c_int32, c_int32, c_int32, c_void_p,
                                         c_void_p]),
    "ali_adam": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, c_float, c_float,
                           c_int32, c_void_p, c_void_p, c_float, c_void_p, c_void_p]),
    "ali_add_i64_multi": (c_int32, [c_int32, POINTER(c_void_p), POINTER(c_int64), c_void_p]),
    "ali_assemble_planes": (c_int32, [c_void_p, c_void_p, POINTER(c_void_p), c_int32, c_void_p, c_int32, c_void_p,
                                      c_int32, c_int32, c_int32, c_int32, c_void_p, c_int32, c_void_p]),
    "ali_spect_post": (c_int32, [c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_float, c_void_p, c_void_p]),
    "ali_plane_table_grad": (c_int32, [c_void_p, c_int32, c_int32, c_void_p, c_int32, c_int32, c_void_p, c_int32, c_int32,
                                       c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p]),
    "ali_col2im": (c_int32, [c_void_p, c_int32, c_void_p, c_void_p] + [c_int32] * 12 + [c_float, c_void_p]),
    "ali_tconv_scatter_ok": (c_int32, [c_int32] * 5),
    "ali_tconv_scatter_wgrad_ws": (c_int64, [c_int32] * 7),
    "ali_tconv_scatter_wgrad": (c_int32, [c_void_p, c_void_p, c_int32, c_void_p, c_int64, c_int64] + [c_int32] * 10
                                + [c_void_p, c_size_t, c_void_p]),
    "ali_tconv_scatter": (c_int32, [c_void_p] * 4 + [c_int32] * 13 + [c_float, c_void_p]),
    "ali_tconv1_fwd": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p] + [c_int32] * 9 + [c_float, c_void_p, c_int32,
                                                                                            c_void_p]),
    "ali_tconv1_dgrad": (c_int32, [c_void_p, c_int32, c_void_p, c_void_p, c_int32, c_float, c_void_p] + [c_int32] * 7
                         + [c_void_p]),
    "ali_tconv1_wgrad": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_int64, c_int64, c_int64]
                         + [c_int32] * 7 + [c_void_p, c_size_t, c_void_p]),
    "ali_last_error": (c_char_p, []),
    "ali_head_fwd": (c_int32, [c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p]),
    "ali_head_wgrad": (c_int32, [c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p]),
    "ali_copy_multi": (c_int32, [c_int32, POINTER(c_void_p), POINTER(c_void_p), POINTER(c_int64), c_void_p]),
    "ali_version": (c_int32, []),
    "ali_reload_tuning": (None, []),
}

_lib = None
_load_error = None


def load():
    """Load libali_hip.so (no GPU needed just to load and resolve symbols)."""
    global _lib, _load_error
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        _load_error = f"{LIB_PATH} not found: build it with `python __graft_entry__.py build` (hipcc, gfx950)"
        raise AliHipUnavailable(_load_error)
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        _load_error = f"cannot load {LIB_PATH}: {e}"
        raise AliHipUnavailable(_load_error)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the ABI and the header drifted apart
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().ali_last_error()
        raise RuntimeError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")
