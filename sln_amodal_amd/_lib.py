"""ctypes loader for libsln_amodal_hip.so (the C-ABI drop-in boundary).

The product path fails loudly when the HIP library is missing: there is no CPU
or eager-PyTorch fallback for the native ops.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (SLN_HIP_LIB: another build of the same library, for A/B runs of compile-time variants -- debug sessions)
LIB_PATH = os.environ.get("SLN_HIP_LIB") or os.path.join(_HERE, "csrc", "libsln_amodal_hip.so")
_lib = None

_p = C.c_void_p
_i = C.c_int
_f = C.c_float

# name -> (restype, argtypes); mirrors include/sln_amodal.h one to one.
SIGNATURES = {
    "sln_abi_version": (_i, []),
    "sln_error_string": (C.c_char_p, [_i]),
    "sln_nms_workspace_bytes": (C.c_size_t, [_i, _i]),
    "sln_nms_f32": (_i, [_p, _i, _i, _p, _f, _i, _p, _p, _p, C.c_size_t, _p]),
    "sln_crop_and_resize_fwd_f32": (_i, [_p, _i, _i, _i, _i, _i, _p, _p, _i, _i, _i, _f, _p, _p, _p]),
    "sln_crop_and_resize_bwd_f32": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p, _p, _p]),
    "sln_label_num_objects_u64": (_i, [_p, _i, C.c_int64, _p, _p]),
    "sln_label_zoom_u64": (_i, [_p, C.c_int64, _p, _p, _p, _i, _i, _i, _p, _p]),
    "sln_label_num_objects_ragged_u64": (_i, [_p, _i, C.c_int64, _p, _p, _p]),
    "sln_label_decode_u64": (_i, [_p, _i, _i, _i, _i, _i, _p, _p]),
    "sln_mask_targets_u64": (_i, [_p, _i, _i, _i, _i, _p, _p, _p, _i, _i, _i, _p, _p]),
    "sln_proposal_decode_f32": (_i, [_p, _p, _p, _p, _i, _i, _i, C.POINTER(_f), _f, _f, _p, _p]),
    "sln_topk_workspace_bytes": (C.c_size_t, [_i, _i, _i]),
    "sln_topk_order_f32": (_i, [_p, _i, _i, C.c_long, C.c_long, _i, _p, _p, C.c_size_t, _p]),
    "sln_maxpool_fwd_f32": (_i, [_p] + [_i] * 10 + [_p, _p, _p]),
    "sln_maxpool_bwd_f32": (_i, [_p, _p] + [_i] * 10 + [_p, _p]),
    "sln_upsample2x_add_f32": (_i, [_p, _p, _i, _i, _i, _i, _p, _p]),
    "sln_sumpool2x2_f32": (_i, [_p, _i, _i, _i, _i, _p, _p]),
    "sln_grad_sqnorm_f32": (_i, [_p, _p, _p, _p, _i, _i, _p, _p, _p]),
    "sln_sgd_clip_step_f32": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _p, _f, _f, _f, _p, _p]),
    "sln_unmold_masks_u8": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _p]),
    "sln_rle_encode_u8": (_i, [_p, _i, C.c_int64, _i, _p, _p, _p]),
    "sln_rle_to_string": (C.c_int64, [_p, C.c_int64, _p, C.c_int64]),
    "sln_rle_from_string": (C.c_int64, [_p, C.c_int64, _p, C.c_int64]),
    "sln_rle_to_strings": (C.c_int64, [_p, C.c_int64, _p, _i, _p, C.c_int64, _p]),
    "sln_gather_rois_f32": (_i, [_p, _p, _p, _i, _i, _i, _f, _f, _p, _p]),
    "sln_pyramid_crop_fwd_f32": (_i, [C.POINTER(_p), C.POINTER(_i), _i, _i, _p, _p, _p, _i, _i, _i,
                                      _f, _p, _i, _i, _p]),
    "sln_pyramid_crop_bwd_f32": (_i, [_p, _i, _i, _p, _p, _p, _i, _i, _i, _i, _i, C.POINTER(_p),
                                      C.POINTER(_i), _i, _p]),
    "sln_pyramid_crop_bwd_gather_workspace_bytes": (C.c_size_t, [_i, _i]),
    "sln_pyramid_crop_bwd_gather_f32": (_i, [_i, C.POINTER(_p), C.POINTER(_i), C.POINTER(_i), C.POINTER(_p),
                                             C.POINTER(_p), C.POINTER(_p), C.POINTER(_i), C.POINTER(_i), C.POINTER(_i),
                                             _i, _i, C.POINTER(_p), C.POINTER(_i), _p, C.c_size_t, _p]),
    "sln_conv_tiled_weight_elems": (C.c_int64, [_i, _i, _i, _i, _i]),
    "sln_conv_fwd_weights_layout": (_i, [C.c_int64, _i, _i, _i, _i, C.c_int64]),
    "sln_debug_read_stamps": (_i, [_p]),
    "sln_conv_split_weights_batch_f32": (_i, [_p, _p, _p, _i, _i, _p]),
    "sln_conv_split_weights_f32": (_i, [_p, _i, _i, _i, _i, _i, C.c_long, C.c_long, C.c_long,
                                        C.c_long, _i, _i, _i, _p, _p, _p, _p, _p]),
    "sln_act_split_f32": (_i, [_p, C.c_int64, _i, _i, _i, _p, _p, _p, _p, _p]),
    "sln_col2im_f32": (_i, [_p] + [_i] * 13 + [_p, _p]),
    "sln_im2col_split_f32": (_i, [_p] + [_i] * 14 + [_p, C.c_int64, C.c_int64, _p, _p, _p, _p]),
    "sln_conv_grad_prep_f32": (_i, [_p, _p, _p, _p, C.c_int64, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p]),
    "sln_scale_update_f32": (_i, [_p, _p, _p, _p, _i, C.c_int64, _i, _i, _p]),
    "sln_scale_update_headroom_f32": (_i, [_p, _p, _p, _p, _p, _i, C.c_int64, _i, _i, _p]),
    "sln_conv2d_fwd_f32": (_i, [_p, _i, _i, _i, _i, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i,
                                _i, _p, _p, _p, _i, _p, _p, _p, _p, _p, _p, _p, _p]),
    "sln_conv2d_fwd_ms_f32": (_i, [_p, _i, _p, _i, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i,
                                   _p, _p, _p, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "sln_conv_fwd_tile": (_i, [C.c_int64, _i, C.c_int64, _i]),
    "sln_conv_fwd_last_kernel": (_i, []),
    "sln_conv_wgrad_last_kernel": (_i, []),
    "sln_conv_wgrad_tile": (_i, [C.c_int64, _i, _i, _i, _i]),
    "sln_conv_wgrad_workspace_bytes": (C.c_size_t, [C.c_int64, _i, _i, _i, _i]),
    "sln_grouped_conv3x3_f32": (_i, [_p, _i, _i, _i, _i, _i, _p, _i, _p, _p, _i, _p, _p]),
    "sln_grouped_conv3x3_dgrad_f32": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p, _i, _p, _p]),
    "sln_grouped_conv3x3_wgrad_workspace_bytes": (C.c_size_t, [_i, _i, _i, _i, _i, _i]),
    "sln_grouped_conv3x3_wgrad_f32": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _p, C.c_size_t, _p]),
    "sln_grouped_conv3x3_packed_weight_elems": (C.c_int64, [_i, _i]),
    "sln_grouped_conv3x3_pack_weights_f16": (_i, [_p, _i, _i, _i, _p, _p, _p, _p, _p]),
    "sln_grouped_conv3x3_f16": (_i, [_p, _i, _i, _i, _i, _i, _p, _i, _i, _p, _p, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "sln_grouped_conv3x3_wgrad_f16": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _p, _p, _p, _p, C.c_size_t, _p]),
    "sln_msc_softmax_tail_f32": (_i, [_p, C.c_int64, _p, _p, _p, _i, _i, _i, _i, _i, _p, _p, _p]),
    "sln_conv_grad_prep_pooled_f32": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p, _p, _i, _i, _p, _p, _p, _p, _p, _p]),
    "sln_conv_wgrad_ksplit": (_i, [C.c_int64, _i, _i, _i, _i]),
    "sln_wgrad_reduce_batch_f32": (_i, [_p, _i, _p]),
    "sln_conv2d_wgrad_f32": (_i, [_p, _i, _i, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i,
                                  _i, _i, _i, _i, _p, _p, _p, _p, C.c_size_t, _i, _p]),
}


class HipExtensionMissing(RuntimeError):
    pass


def register(name, restype, argtypes):
    SIGNATURES[name] = (restype, argtypes)
    if _lib is not None:
        fn = getattr(_lib, name)
        fn.restype, fn.argtypes = restype, argtypes


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipExtensionMissing(
                "%s not found: build it with `python -m sln_amodal_amd.csrc.build` "
                "(hipcc --offload-arch=gfx950). No CPU fallback exists." % LIB_PATH)
        # Load PyTorch's HIP runtime FIRST: the library's DT_NEEDED libamdhip64.so.7
        # then binds to the copy already in the process (same SONAME).  Loading ours
        # first would pull /opt/rocm's runtime in and hand torch a second, different
        # HIP runtime ("no ROCm-capable device", mismatched stream handles).
        import torch  # noqa: F401
        _lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(_lib, name)  # AttributeError if the .so lacks a declared symbol
            fn.restype, fn.argtypes = res, args
    return _lib


def check(code, what):
    if code != 0:
        raise RuntimeError("%s failed: %s" % (what, lib().sln_error_string(code).decode()))
