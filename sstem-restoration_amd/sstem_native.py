"""Single loader of ``csrc/libsstem_hip.so`` and the ctypes prototypes of its whole C-ABI
(``include/sstem_sepconv.h``, ``include/sstem_conv.h``, ``include/sstem_warp.h``, ``include/sstem_resize.h``, ``include/sstem_norm.h``, ``include/sstem_io.h``).  No fallback: a missing library raises."""
import ctypes
import os

_PKG_ROOT = os.path.dirname(os.path.abspath(__file__))
# SSTEM_NATIVE_LIB: developer override (ablation builds); the product library is csrc/libsstem_hip.so
_LIB_PATH = os.environ.get("SSTEM_NATIVE_LIB") or os.path.join(_PKG_ROOT, "csrc", "libsstem_hip.so")
_lib = None

_p = ctypes.c_void_p
_i64 = ctypes.c_int64
_int = ctypes.c_int
_f = ctypes.c_float

# name -> (restype, argtypes); exactly the prototypes of include/*.h
C_ABI = {
    # include/sstem_sepconv.h
    "sstem_sepconv_forward_f32": (_int, [_p] * 4 + [_i64] * 4 + [_p]),
    "sstem_sepconv_forward_f32_algo": (_int, [_p] * 4 + [_i64] * 4 + [_p, _int]),
    "sstem_sepconv_backward_f32": (_int, [_p] * 7 + [_i64] * 4 + [_p]),
    "sstem_sepconv_backward_f32_algo": (_int, [_p] * 7 + [_i64] * 4 + [_p, _int]),
    "sstem_sepconv_interp_apply_f32": (_int, [_p] * 7 + [_i64] * 3 + [_p]),
    "sstem_sepconv_interp_apply_gray_f32": (_int, [_p] * 7 + [_i64] * 3 + [_p]),
    "sstem_sepconv_interp_apply_gray_supported": (_int, [_i64] * 3),
    "sstem_sepconv_coef_blocked_floats": (_i64, [_i64] * 3),
    "sstem_sepconv_coef_to_blocked_f32": (_int, [_p, _p] + [_i64] * 3 + [_p]),
    "sstem_sepconv_interp_apply_gray_blocked_f32": (_int, [_p] * 7 + [_i64] * 3 + [_p]),
    "sstem_sepconv_interp_apply_gray_blocked_supported": (_int, [_i64] * 3),
    "sstem_sepconv_interp_apply_gray_u8_f32": (_int, [_p] * 8 + [_i64] * 3 + [_int, _p]),
    "sstem_sepconv_interp_apply_bytes": (_i64, [_i64] * 3 + [_int]),
    "sstem_sepconv_forward_bytes": (_i64, [_i64] * 4),
    "sstem_sepconv_backward_bytes": (_i64, [_i64] * 4),
    "sstem_sepconv_forward_taps_f32": (_int, [_p] * 4 + [_i64] * 4 + [_int, _p]),
    "sstem_sepconv_backward_taps_f32": (_int, [_p] * 6 + [_i64] * 4 + [_int, _p]),
    "sstem_sepconv_forward_bf16coef": (_int, [_p] * 4 + [_i64] * 4 + [_p]),
    "sstem_sepconv_backward_bf16coef": (_int, [_p] * 7 + [_i64] * 4 + [_p]),
    "sstem_sepconv_interp_apply_gray_bf16coef": (_int, [_p] * 7 + [_i64] * 3 + [_p]),
    "sstem_sepconv_interp_apply_gray_bf16coef_supported": (_int, [_i64] * 3),
    "sstem_sepconv_forward_bytes_bf16coef": (_i64, [_i64] * 4),
    "sstem_sepconv_backward_bytes_bf16coef": (_i64, [_i64] * 4),
    "sstem_sepconv_interp_apply_bytes_bf16coef": (_i64, [_i64] * 3 + [_int]),
    "sstem_version": (_int, []),
    "sstem_status_string": (ctypes.c_char_p, [_int]),
    "sstem_last_error": (ctypes.c_char_p, []),
    # include/sstem_conv.h
    "sstem_conv3x3_workspace_floats": (_i64, [_i64, _i64]),
    "sstem_conv3x3_forward_workspace_floats": (_i64, [_i64] * 5),
    "sstem_conv3x3_forward_workspace_floats_algo": (_i64, [_i64] * 5 + [_int]),
    "sstem_conv3x3_packed_floats": (_i64, [_i64, _i64, _int]),
    "sstem_conv3x3_pack_weights_f32": (_int, [_p, _i64, _i64, _int, _p, _p, _p]),
    "sstem_conv3x3_pack_group_entry": (_i64, [_i64, _i64, _int, _p]),
    "sstem_conv3x3_pack_weights_group_f32": (_int, [_p, _i64, _i64, _int, _p]),
    "sstem_conv3x3_pack_weights_group_f16": (_int, [_p, _i64, _i64, _i64, _p, _p]),
    "sstem_conv2d_forward_f32": (_int, [_p] * 7 + [_i64] + [_i64] * 5 + [_int] * 5 + [_int, _f, _p, _int]),
    "sstem_conv_bn_partials": (_i64, [_i64] * 5 + [_int] * 4),
    "sstem_conv2d_forward_ex_f32": (_int, [_p] * 6 + [_f] + [_p] * 3 + [_i64] + [_i64] * 5 + [_int] * 5 + [_int, _f, _p, _int]),
    "sstem_conv_transpose3x3s2_workspace_floats": (_i64, [_i64] * 5 + [_int]),
    "sstem_conv_transpose3x3s2_forward_ex_f32": (_int, [_p] * 6 + [_f] + [_p] * 3 + [_i64] + [_i64] * 5 + [_int, _int, _f, _p]),
    "sstem_conv_transpose3x3s2_backward_ex_f32": (_int, [_p] * 7 + [_i64] + [_i64] * 5 + [_int, _p]),
    "sstem_conv2d_backward_weight_bias_ex_f32": (_int, [_p] * 5 + [_i64] + [_i64] * 5 + [_int] * 5 + [_p, _int]),
    "sstem_conv3x3_backward_weight_bf16in_ex": (_int, [_p] * 5 + [_i64] + [_i64] * 5 + [_int, _p]),
    "sstem_conv3x3_algo_supported": (_int, [_i64] * 5 + [_int]),
    "sstem_conv3x3_forward_bf16io_masked": (_int, [_p, _int, _p, _p, _p, _p, _p, _p, _int, _p, _p, _i64] + [_i64] * 5 + [_int, _int, _f, _p]),
    "sstem_conv3x3_backward_weight_bf16_masked": (_int, [_p, _int, _p, _p, _p, _p, _p, _i64] + [_i64] * 5 + [_int, _p]),
    "sstem_conv3x3_forward_masked_f32": (_int, [_p] * 9 + [_i64] + [_i64] * 5 + [_int, _int, _f, _p, _int]),
    "sstem_conv3x3_backward_weight_masked_f32": (_int, [_p] * 6 + [_i64] + [_i64] * 5 + [_int, _p, _int]),
    "sstem_amax_word_floats": (_i64, []),
    "sstem_amax_f32": (_int, [_p, _i64, _p, _p]),
    "sstem_conv3x3_forward_scaled_f32": (_int, [_p] * 7 + [_f] + [_p] * 3 + [_i64] + [_i64] * 5 + [_int, _int, _f, _p, _int, _int]),
    "sstem_conv3x3_forward_scaled_strided_f32": (_int, [_p] * 7 + [_f] + [_p] * 3 + [_i64] + [_i64] * 5 + [_int, _int, _f, _p, _int, _int, _i64, _p, _int]),
    "sstem_conv3x3_bf16io_supported": (_int, [_i64] * 5 + [_int]),
    "sstem_conv3x3_stream_small_supported": (_int, [_i64] * 5),
    "sstem_conv3x3_forward_scaled_masked_f32": (_int, [_p] * 11 + [_i64] + [_i64] * 5 + [_int, _int, _f, _p]),
    "sstem_conv3x3_backward_weight_scaled_masked_f32": (_int, [_p] * 8 + [_i64] + [_i64] * 5 + [_int, _p]),
    "sstem_wgrad_deferred_count": (_int, []),
    "sstem_wgrad_deferred_drop": (None, []),
    "sstem_wgrad_deferred_flush": (_int, [_p]),
    "sstem_conv3x3_first_layer_u8_supported": (_int, [_i64] * 4),
    "sstem_conv3x3_first_layer_u8": (_int, [_p] * 6 + [_i64] * 4 + [_int, _f, _p]),
    "sstem_conv3x3_forward_bf16io": (_int, [_p, _int, _p, _p, _p, _p, _p, _int, _p, _i64] + [_i64] * 5 + [_int, _int, _f, _p]),
    "sstem_conv_transpose3x3s2_forward_f32": (_int, [_p] * 6 + [_i64] * 5 + [_int, _f, _p]),
    "sstem_conv2d_backward_weight_f32": (_int, [_p] * 4 + [_i64] + [_i64] * 5 + [_int] * 4 + [_p, _int]),
    "sstem_conv2d_backward_weight_bias_f32": (_int, [_p] * 5 + [_i64] + [_i64] * 5 + [_int] * 4 + [_p, _int]),
    "sstem_conv3x3_wgrad_workspace_floats": (_i64, [_i64] * 5),
    "sstem_conv3x3_wgrad_workspace_floats_algo": (_i64, [_i64] * 5 + [_int]),
    "sstem_conv3x3_backward_weight_bf16in": (_int, [_p] * 5 + [_i64] + [_i64] * 5 + [_p]),
    "sstem_conv_transpose3x3s2_backward_f32": (_int, [_p] * 5 + [_i64] * 5 + [_p]),
    # include/sstem_warp.h
    "sstem_warp_bilinear_f32": (_int, [_p] * 3 + [_i64] * 4 + [_p]),
    # include/sstem_norm.h
    "sstem_batchnorm_workspace_floats": (_i64, [_i64] * 3),
    "sstem_batchnorm_train_forward_f32": (_int, [_p] * 9 + [_i64] + [_i64] * 3 + [_f, _f, _int, _f, _p]),
    "sstem_batchnorm_train_forward_ex_f32": (_int, [_p] * 10 + [_i64, _p, _i64] + [_i64] * 3 + [_f, _f, _int, _f, _p]),
    "sstem_batchnorm_train_backward_ex_f32": (_int, [_p] * 10 + [_i64] + [_i64] * 3 + [_int, _f, _int, _p]),
    "sstem_batchnorm_train_forward_amax_f32": (_int, [_p] * 11 + [_i64, _p, _i64] + [_i64] * 3 + [_f, _f, _int, _f, _p]),
    "sstem_batchnorm_train_backward_amax_f32": (_int, [_p] * 11 + [_i64] + [_i64] * 3 + [_int, _f, _int, _p]),
    "sstem_batchnorm_train_backward_f32": (_int, [_p] * 10 + [_i64] + [_i64] * 3 + [_int, _f, _p]),
    # include/sstem_resize.h
    "sstem_upsample_bilinear2x_f32": (_int, [_p, _p, _i64, _i64, _i64, _p]),
    "sstem_upsample_bilinear2x_backward_f32": (_int, [_p, _p, _i64, _i64, _i64, _p]),
    "sstem_pool2x2_forward_f32": (_int, [_p, _p, _p, _i64, _i64, _i64, _int, _p]),
    "sstem_pool2x2_backward_f32": (_int, [_p, _p, _p, _i64, _i64, _i64, _int, _p]),
    # include/sstem_io.h
    "sstem_gray_u8_to_f32": (_int, [_p, _p, _i64, _i64, _p]),
    "sstem_f32_to_gray_u8": (_int, [_p, _p, _i64, _int, _p]),
    "sstem_adam_step_f32": (_int, [_p] * 4 + [_i64] + [_f] * 5 + [_i64, _p]),
    "sstem_l1_workspace_floats": (_i64, []),
    "sstem_l1_mean_forward_grad_f32": (_int, [_p, _p, _i64, _p, _p, _p, _p]),
}


def library_path():
    return _LIB_PATH


def load_library():
    """Load libsstem_hip.so once; raise loudly when it is not there."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise ImportError(
                "libsstem_hip.so not found at %s -- build it with "
                "`python -c 'import __graft_entry__ as g; g.build()'` or `make -C %s`. "
                "There is no CPU/PyTorch fallback for the native ops."
                % (_LIB_PATH, os.path.dirname(_LIB_PATH)))
        lib = ctypes.CDLL(_LIB_PATH)
        for name, (res, args) in C_ABI.items():
            fn = getattr(lib, name)  # AttributeError if the library does not export it
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(rc, what):
    if rc != 0:
        lib = load_library()
        detail = lib.sstem_last_error().decode("utf-8", "replace")
        name = lib.sstem_status_string(rc).decode("utf-8", "replace")
        raise RuntimeError("%s failed: %s (%d)%s" % (what, name, rc, (": " + detail) if detail else ""))
