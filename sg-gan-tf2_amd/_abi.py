"""ctypes binding of include/sggan.h (libsggan.so).

There is NO fallback: if the HIP library is missing or a symbol is absent this
module raises at import/first use -- the product path never routes through a
CPU implementation (the CPU oracle under oracle/ is test infrastructure only).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsggan.so")      # the package reads no environment; tools pick another build with use_library()

SGG_F32, SGG_BF16 = 0, 1
ACT_NONE, ACT_RELU, ACT_LRELU, ACT_TANH = 0, 1, 2, 3
PAD_ZERO, PAD_REFLECT = 0, 1
CPAD = 8


class ConvDesc(C.Structure):
    """struct sgg_conv_desc (include/sggan.h)."""
    _fields_ = [(n, C.c_int32) for n in
                ("N", "H", "W", "C", "K", "R", "S", "stride", "pad_t", "pad_l", "Ho", "Wo", "pad_mode", "dtype")]


class PackItem(C.Structure):
    """struct sgg_pack_item (include/sggan.h)."""
    _fields_ = [("w", C.c_void_p), ("w_fwd", C.c_void_p), ("w_dgrad", C.c_void_p)] + [(n, C.c_int32) for n in
                ("taps", "C", "K", "Cpad", "Kpad", "reserved")]


_vp, _i, _i64, _f, _sz = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_size_t
_dp = C.POINTER(ConvDesc)

# name -> (restype, argtypes); every symbol include/sggan.h declares
SIGNATURES = {
    "sgg_version": (_i, []),
    "sgg_strerror": (C.c_char_p, [_i]),
    "sgg_event_create": (_i, [C.POINTER(C.c_void_p)]),
    "sgg_event_destroy": (_i, [_vp]),
    "sgg_event_elapsed_ms": (_i, [_vp, _vp, C.POINTER(C.c_float)]),
    "sgg_time_next_launch": (_i, [_vp, _vp]),
    "sgg_stream_capture_nodes": (_i, [_vp, C.POINTER(C.c_int)]),
    "sgg_pack_conv_weights": (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "sgg_pack_conv_weights_batch": (_i, [_vp, _i, _i64, _i, _vp]),
    "sgg_conv2d_fwd_workspace": (_sz, [_dp]),
    "sgg_conv2d_fwd": (_i, [_dp, _vp, _vp, _vp, _vp, _i, _f, _vp, _sz, _vp]),
    "sgg_conv2d_fwd_stats_chunks": (_sz, [_dp]),
    "sgg_conv2d_fwd_stats": (_i, [_dp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "sgg_conv2d_bwd_data_workspace": (_sz, [_dp]),
    "sgg_conv2d_bwd_data": (_i, [_dp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "sgg_conv2d_bwd_data_stats_chunks": (_sz, [_dp]),
    "sgg_conv2d_bwd_data_stats": (_i, [_dp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _f, _vp, _vp, _sz, _vp]),
    "sgg_conv2d_bwd_data_mixed_supported": (_i, [_dp]),
    "sgg_conv2d_bwd_data_mixed": (_i, [_dp, _vp, _vp, _vp, _i, _vp, _vp, _sz, _vp]),
    "sgg_conv2d_pair_supported": (_i, [_dp]),
    "sgg_conv2d_fwd_stats_pair": (_i, [_dp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _sz, _vp]),
    "sgg_conv2d_fwd_normload_supported": (_i, [_dp]),
    "sgg_conv2d_fwd_stats_normload": (_i, [_dp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _sz, _vp]),
    "sgg_conv2d_bwd_data_pair": (_i, [_dp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _sz, _vp]),
    "sgg_conv2d_fwd_group2": (_i, [_dp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _f, _vp, _sz, _vp]),
    "sgg_conv2d_bwd_data_group2": (_i, [_dp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "sgg_deconv2d_fwd_group2": (_i, [_dp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _f, _vp, _sz, _vp]),
    "sgg_deconv2d_bwd_data_group2": (_i, [_dp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "sgg_conv2d_bwd_weight_group2": (_i, [_dp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _sz, _vp]),
    "sgg_deconv2d_bwd_weight_group2": (_i, [_dp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _sz, _vp]),
    "sgg_conv2d_bwd_weight_workspace": (_sz, [_dp]),
    "sgg_conv2d_bwd_weight": (_i, [_dp, _vp, _vp, _vp, _i, _i, _i, _vp, _sz, _vp]),
    "sgg_conv2d_bwd_weight_pair_supported": (_i, [_dp]),
    "sgg_conv2d_bwd_weight_pair": (_i, [_dp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _sz, _vp]),
    "sgg_conv2d_bwd_weight_pair2": (_i, [_dp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _sz, _vp]),
    "sgg_deconv2d_fwd_workspace": (_sz, [_dp]),
    "sgg_deconv2d_fwd": (_i, [_dp, _vp, _vp, _vp, _vp, _i, _f, _vp, _sz, _vp]),
    "sgg_deconv2d_fwd_stats_chunks": (_sz, [_dp]),
    "sgg_deconv2d_fwd_stats": (_i, [_dp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _sz, _vp]),
    "sgg_deconv2d_bwd_data_workspace": (_sz, [_dp]),
    "sgg_deconv2d_bwd_data": (_i, [_dp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "sgg_deconv2d_bwd_weight": (_i, [_dp, _vp, _vp, _vp, _i, _i, _i, _vp, _sz, _vp]),
    "sgg_bias_grad_workspace": (_sz, [_i64, _i]),
    "sgg_bias_grad": (_i, [_vp, _vp, _i64, _i, _i, _i, _i, _vp, _sz, _vp]),
    "sgg_bias_grad_group2": (_i, [_vp, _vp, _vp, _i64, _i, _i, _i, _i, _vp, _sz, _vp]),
    "sgg_instnorm_workspace": (_sz, [_i, _i64, _i]),
    "sgg_instnorm_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i64, _i, _f, _i, _f, _i, _vp, _sz, _vp]),
    "sgg_instnorm_fwd_partial": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i64, _i, _f, _i, _f, _i, _vp]),
    "sgg_instnorm_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i64, _i, _i, _i, _i, _f, _i, _vp, _sz, _vp]),
    "sgg_instnorm_finalize": (_i, [_vp, _i, _vp, _i, _i64, _i, _f, _vp]),
    "sgg_instnorm_fwd_pair": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _i64, _i, _f, _i, _f, _i, _vp, _sz, _vp]),
    "sgg_instnorm_fwd_partial_pair": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i64, _i, _f, _i, _f, _i, _vp]),
    "sgg_instnorm_bwd_pair": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i64, _i, _i, _i, _i, _f, _i, _vp, _sz, _vp]),
    "sgg_instnorm_bwd_mixed": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i64, _i, _i, _i, _i, _f, _vp, _sz, _vp]),
    "sgg_instnorm_bwd_partial": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i64, _i, _i, _i, _i, _f, _i, _vp, _sz, _vp]),
    "sgg_act_fwd": (_i, [_vp, _vp, _i64, _i, _f, _i, _vp]),
    "sgg_act_bwd": (_i, [_vp, _vp, _vp, _i64, _i, _f, _i, _vp]),
    "sgg_add": (_i, [_vp, _vp, _vp, _i64, _i, _vp]),
    "sgg_mask_reduce_fwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "sgg_mask_reduce_bwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "sgg_bce_logits": (_i, [_vp, _i64, _f, _f, _f, _vp, _vp, _i, _vp]),
    "sgg_l1_loss_workspace": (_sz, [_i64, _i]),
    "sgg_l1_loss": (_i, [_vp, _vp, _i64, _i, _i, _f, _f, _vp, _vp, _i, _i, _vp, _sz, _vp]),
    "sgg_mse_const": (_i, [_vp, _i64, _f, _f, _f, _vp, _vp, _i, _vp]),
    "sgg_seg_edge_weight": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "sgg_gradloss_workspace": (_sz, [_i, _i, _i, _i]),
    "sgg_gradloss": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _f, _vp, _vp, _i, _i, _vp, _sz, _vp]),
    "sgg_adam": (_i, [_vp, _vp, _vp, _vp, _i64, _i, _f, _f, _f, _f, _f, _vp]),
    "sgg_adam_iter": (_i, [_vp, _vp, _vp, _vp, _i64, _vp, _f, _f, _f, _f, _f, _vp]),
    "sgg_seg_class_map": (_i, [_vp, _i, _i64, _vp, _vp]),
    "sgg_seg_class_table": (_i, [_vp, _vp, _i]),
    "sgg_onehot_resample": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "sgg_confusion_hist": (_i, [_vp, _vp, _i64, _i, _vp, _vp]),
    "sgg_argmax_u8_labels": (_i, [_vp, _vp, _i64, _i, _i, _i, _vp]),
    "sgg_pad_channels": (_i, [_vp, _vp, _i64, _i, _i, _i, _vp]),
    "sgg_unpad_channels": (_i, [_vp, _vp, _i64, _i, _i, _i, _vp]),
}

_lib = None


def use_library(path):
    """Bind another build of the library (an A/B variant or the -DSGG_LAB build that tools/ use) instead of the in-tree
    libsggan.so.  Must be called before the first kernel call; the training path never calls it."""
    global LIB_PATH
    if _lib is not None:
        raise RuntimeError("use_library() after the library was loaded")
    LIB_PATH = os.path.abspath(path)


def lib():
    """Load libsggan.so (once) and bind every declared symbol; raises if anything is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: the HIP extension is not built. Run `python __graft_entry__.py build` "
                "(there is deliberately no CPU fallback).")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)          # AttributeError if the .so does not export it
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


class SggError(RuntimeError):
    pass


OK, EINVAL, EUNSUPPORTED, ELAUNCH, EWORKSPACE = 0, -1, -2, -3, -4      # sgg_status (include/sggan.h)


def check(status: int, what: str = ""):
    if status != 0:
        raise SggError(f"{what}: {lib().sgg_strerror(status).decode()} ({status})")
