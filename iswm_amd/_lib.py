"""ctypes binding of libiswm_hip.so (the C ABI declared in include/iswm_hip.h).

The product path has NO fallback: if the shared library is missing or a symbol
cannot be resolved, importing the ops raises -- nothing silently routes through
torch ops or the CPU oracle.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_int64, c_size_t, c_uint64, c_void_p

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libiswm_hip.so")


class ConvDesc(Structure):
    """iswm_conv_desc"""
    _fields_ = [(n, c_int) for n in ("N", "H", "W", "Cin", "Ho", "Wo", "Cout", "KH", "KW", "stride", "pad",
                                     "dil", "ldx", "ldy")]


P = c_void_p
_SIGS = {
    # name: (restype, [argtypes])
    "iswm_last_error": (c_char_p, []),
    "iswm_version": (c_int, []),
    "iswm_set_conv_math": (c_int, [c_int]),
    "iswm_get_conv_math": (c_int, []),
    "iswm_conv2d_kernel_name": (c_int, [POINTER(ConvDesc), c_int, c_char_p, c_int]),
    "iswm_conv2d_stat_tile_rows": (c_int, [POINTER(ConvDesc)]),
    "iswm_conv2d_stat_tiles": (c_int, [POINTER(ConvDesc)]),
    "iswm_conv2d_fwd": (c_int, [POINTER(ConvDesc), P, P, P, P, P, P]),
    "iswm_conv2d_dgrad": (c_int, [POINTER(ConvDesc), P, P, P, c_int, P]),
    "iswm_transpose_weights": (c_int, [POINTER(ConvDesc), P, P, P]),
    "iswm_conv2d_dgrad_wants_wt": (c_int, [POINTER(ConvDesc)]),
    "iswm_conv2d_dgrad_wt": (c_int, [POINTER(ConvDesc), P, P, P, c_int, P]),
    "iswm_conv2d_packed_weight_bytes": (c_size_t, [POINTER(ConvDesc), c_int]),
    "iswm_conv2d_pack_weights": (c_int, [POINTER(ConvDesc), c_int, P, P, P]),
    "iswm_packed_weight_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "iswm_pack_job_blocks": (c_int, [c_int, c_int, c_int, c_int]),
    "iswm_pack_weights_batch": (c_int, [P, c_int, c_int, P]),
    "iswm_conv2d_fwd_packed_stat_layout": (c_int, [POINTER(ConvDesc), POINTER(c_int), POINTER(c_int)]),
    "iswm_conv2d_fwd_packed": (c_int, [POINTER(ConvDesc), P, P, P, P, P, P]),
    "iswm_conv2d_dgrad_packed": (c_int, [POINTER(ConvDesc), P, P, P, c_int, P]),
    "iswm_split_planes": (c_int, [P, c_int64, c_int, c_int, P, c_int, c_int64, P]),
    "iswm_join_planes": (c_int, [P, c_int, c_int64, c_int64, c_int, P, c_int, P]),
    "iswm_bn_apply_pl": (c_int, [P, c_int64, c_int, c_int, P, P, P, P, c_int, c_int64, c_int, P, c_int, c_int64, P]),
    "iswm_bn_backward_pl": (c_int, [P, c_int, P, c_int, c_int64, P, c_int, c_int64, c_int, P, P, P, P, P, c_int, c_int, P, P, P,
                                    c_int, c_int64, P, c_int, P, c_size_t, P]),
    "iswm_maxpool3x3s2_fwd_pl": (c_int, [P, c_int, c_int, c_int, c_int, P, c_int64, P, c_int, c_int, P]),
    "iswm_gap_fwd_pl": (c_int, [P, c_int64, c_int, c_int, c_int, c_int, P, P]),
    "iswm_bcast_fwd_pl": (c_int, [P, c_int, c_int, c_int, P, c_int, c_int64, P]),
    "iswm_bilinear_fwd_pl": (c_int, [P, c_int, c_int, c_int, c_int, c_int, P, c_int64, c_int, c_int, c_int, P]),
    "iswm_set_debug_buffer": (c_int, [P]),
    "iswm_conv2d_pl2_weight_bytes": (c_size_t, [POINTER(ConvDesc), c_int]),
    "iswm_conv2d_pl2_pack_weights": (c_int, [POINTER(ConvDesc), c_int, P, P, P]),
    "iswm_conv2d_pl2_tile_rows": (c_int, [POINTER(ConvDesc), c_int]),
    "iswm_conv2d_fwd_pl2": (c_int, [POINTER(ConvDesc), P, c_int64, P, P, P, P, P]),
    "iswm_conv2d_dgrad_pl2": (c_int, [POINTER(ConvDesc), P, c_int64, P, P, c_int, P]),
    "iswm_conv2d_dgrad_pl2_stat_tiles": (c_int, [POINTER(ConvDesc)]),
    "iswm_conv2d_dgrad_pl2_bn": (c_int, [POINTER(ConvDesc), P, c_int64, P, P, c_int, P, c_int, P, P, P, P, c_int, P, c_int, P,
                                         c_int, P]),
    "iswm_bn_backward_stats_pl": (c_int, [P, c_int, P, c_int, c_int64, P, c_int, c_int64, c_int, P, P, P, P, P, c_int, c_int, P, P,
                                          P, c_int, c_int64, P, c_int, P, c_int, P, c_size_t, P]),
    "iswm_aspp_plan_bytes": (c_size_t, [POINTER(ConvDesc), c_int, POINTER(c_int), POINTER(c_int), c_int]),
    "iswm_aspp_plan": (c_int, [POINTER(ConvDesc), c_int, POINTER(c_int), POINTER(c_int), c_int, P, c_int]),
    "iswm_aspp_fwd": (c_int, [POINTER(ConvDesc), c_int, POINTER(c_int), POINTER(c_int), P, P, c_int64, POINTER(c_void_p),
                              POINTER(c_void_p), POINTER(c_void_p), P]),
    "iswm_aspp_bwd": (c_int, [POINTER(ConvDesc), c_int, POINTER(c_int), POINTER(c_int), P, P, c_int64, c_int, POINTER(c_void_p), P,
                              c_int, P, c_int64, POINTER(c_void_p), P, c_size_t, P]),
    "iswm_conv2d_wgrad_planes_ok": (c_int, [POINTER(ConvDesc)]),
    "iswm_conv2d_wgrad_planes_workspace": (c_size_t, [POINTER(ConvDesc)]),
    "iswm_conv2d_wgrad_planes": (c_int, [POINTER(ConvDesc), P, c_int64, P, c_int64, P, P, c_size_t, P]),
    "iswm_conv2d_wgrad_workspace": (c_size_t, [POINTER(ConvDesc)]),
    "iswm_dwconv2d_fwd": (c_int, [POINTER(ConvDesc), P, P, c_int, P, P, P]),
    "iswm_dwconv2d_dgrad": (c_int, [POINTER(ConvDesc), P, P, c_int, P, c_int, P]),
    "iswm_dwconv2d_wgrad_workspace": (c_size_t, [POINTER(ConvDesc)]),
    "iswm_dwconv2d_wgrad": (c_int, [POINTER(ConvDesc), P, P, c_int, P, P, c_size_t, P]),
    "iswm_conv2d_wgrad": (c_int, [POINTER(ConvDesc), P, P, P, P, c_size_t, P]),
    "iswm_colstat_tiles": (c_int, [c_int64]),
    "iswm_colstat_tile_rows": (c_int64, [c_int64]),
    "iswm_colstat": (c_int, [P, c_int64, c_int, c_int, P, P]),
    "iswm_bn_finalize": (c_int, [P, c_int, c_int, c_int64, c_int64, P, P, P, P, c_float, c_float, P, P, P, P, P]),
    "iswm_bn_eval_coeffs": (c_int, [c_int, P, P, P, P, c_float, P, P, P, P, P]),
    "iswm_bn_apply": (c_int, [P, c_int64, c_int, c_int, P, P, P, P, c_int, c_int, P, c_int, P]),
    "iswm_bn_bwd_workspace": (c_size_t, [c_int64, c_int]),
    "iswm_bn_backward": (c_int, [P, c_int, P, c_int, P, c_int, c_int64, c_int, P, P, P, P, P, c_int, c_int, P, P, P, c_int,
                                 P, c_int, P, c_size_t, P]),
    "iswm_colsum_finalize": (c_int, [P, c_int, c_int, P, P, P]),
    "iswm_maxpool3x3s2_fwd": (c_int, [P, c_int, c_int, c_int, c_int, P, P, c_int, c_int, P]),
    "iswm_maxpool3x3s2_bwd": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, c_int, P, P]),
    "iswm_gap_fwd": (c_int, [P, c_int, c_int, c_int, c_int, P, P]),
    "iswm_gap_bwd": (c_int, [P, c_int, c_int, c_int, P, c_int, c_int, P]),
    "iswm_bcast_fwd": (c_int, [P, c_int, c_int, c_int, P, c_int, P]),
    "iswm_bcast_bwd": (c_int, [P, c_int, c_int, c_int, c_int, P, P]),
    "iswm_bilinear_fwd": (c_int, [P, c_int, c_int, c_int, c_int, c_int, P, c_int, c_int, c_int, P]),
    "iswm_bilinear_bwd": (c_int, [P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, c_int, P]),
    "iswm_bilinear_nhwc_to_nchw_fwd": (c_int, [P, c_int, c_int, c_int, c_int, c_int, P, c_int, c_int, P]),
    "iswm_bilinear_nhwc_to_nchw_bwd": (c_int, [P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P]),
    "iswm_nchw_to_nhwc": (c_int, [P, c_int, c_int, c_int, P, c_int, P]),
    "iswm_nhwc_to_nchw": (c_int, [P, c_int, c_int, c_int, c_int, P, P]),
    "iswm_copy_channels": (c_int, [P, c_int, P, c_int, c_int64, c_int, P]),
    "iswm_add_inplace": (c_int, [P, P, c_int64, P]),
    "iswm_scale_inplace": (c_int, [P, c_int64, P, c_float, P]),
    "iswm_bn_apply_classify": (c_int, [P, c_int64, c_int, c_int, P, P, P, P, P, P, c_int, P]),
    "iswm_bn_classify_bwd_workspace": (c_size_t, [c_int64, c_int]),
    "iswm_bn_backward_classify": (c_int, [P, c_int, P, P, c_int, c_int64, c_int, P, P, P, P, P, c_int, P, P, P, P, c_int, c_int64,
                                          P, c_size_t, P]),
    "iswm_pad_weights": (c_int, [P, c_int, c_int, c_int, c_int, P, c_int, c_int, P, P]),
    "iswm_unpad_weights": (c_int, [P, c_int, c_int, c_int, c_int, c_int, c_int, P, P, P]),
    "iswm_fill_zero": (c_int, [P, c_size_t, P]),
    "iswm_zero_cols": (c_int, [P, c_int64, c_int64, c_int, c_int, c_int64, c_int, P]),
    "iswm_dropout_fwd": (c_int, [P, P, P, c_int64, c_float, c_uint64, c_uint64, P]),
    "iswm_dropout_bwd": (c_int, [P, P, P, c_int64, c_float, P]),
    "iswm_loss_blocks": (c_int, [c_int64]),
    "iswm_loss_fwd": (c_int, [P, P, c_int, c_int, c_int, c_int64, P, c_int, c_float, c_float, c_int, P, P, P]),
    "iswm_loss_finalize": (c_int, [P, c_int, c_int, c_int64, P, P, P]),
    "iswm_loss_bwd_scale": (c_int, [P, c_int64, P, P, c_int, c_int64, P]),
    "iswm_argmax_nchw": (c_int, [P, c_int, c_int, c_int64, P, P]),
    "iswm_augment_batch": (c_int, [P, P, P, P, c_int, c_int, c_int, POINTER(c_float), POINTER(c_float), P, P, P]),
    "iswm_confusion_matrix": (c_int, [P, c_int, P, c_int, c_int64, c_int, P, P]),
    "iswm_confusion_matrix_logits": (c_int, [P, c_int, P, c_int, c_int, c_int64, c_int, P, P]),
    "iswm_sgd_step": (c_int, [P, P, P, c_int64, P, c_float, c_float, c_int, P]),
    "iswm_adam_step": (c_int, [P, P, P, P, c_int64, P, c_float, c_float, c_float, c_float, c_int, P]),
}

EXPORTS = tuple(_SIGS)
_lib = None


class IswmError(RuntimeError):
    pass


def load():
    """Load the library once; raise loudly when it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libiswm_hip.so is not built (%s). Run `python -m iswm_amd.build` (hipcc, gfx950). "
            "iswm_amd has no CPU or torch-op fallback." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def call(name, *args):
    """Call a status-returning entry point; non-zero status -> IswmError(message)."""
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        msg = lib.iswm_last_error()
        raise IswmError("%s failed (%d): %s" % (name, rc, msg.decode() if msg else "?"))
