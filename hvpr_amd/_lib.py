"""ctypes binding of libhvpr_amd.so (the C-ABI declared in include/hvpr_amd.h).

The product path has NO CPU fallback: if the library is missing or fails to load, every op raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
ABI_VERSION = 6          # hvpr_abi_version() of the library these wrappers were written against (csrc/abi.hip)
LIB_PATH = os.environ.get("HVPR_AMD_LIB", os.path.join(_HERE, "libhvpr_amd.so"))   # override: kernel experiments only

_c = ctypes
_P = _c.c_void_p
_I = _c.c_int
_F = _c.c_float
_Z = _c.c_size_t
ALLREDUCE_FN = _c.CFUNCTYPE(_I, _P, _I, _P, _P)      # hvpr_allreduce_fn: (double *buf, int n, hvpr_stream_t stream, void *ctx) -> int

# name -> (restype, argtypes); mirrors include/hvpr_amd.h one to one (checked by tests/test_capi_symbols.py)
SIGNATURES = {
    "hvpr_abi_version": (_I, []),
    "hvpr_status_string": (_c.c_char_p, [_I]),
    "hvpr_set_batchnorm_allreduce": (None, [_P, _P]),
    "hvpr_voxelize_workspace_bytes": (_Z, [_I, _I, _I, _I, _I]),
    "hvpr_voxelize_workspace_reset": (_I, [_P, _Z, _I, _I, _I, _I, _I, _P]),
    "hvpr_voxelize_workspace_status": (_I, [_P, _Z, _I, _I, _I, _I, _I, _P]),
    "hvpr_voxelize_f32": (_I, [_P, _I, _I, _I, _I, _P, _I, _F, _F, _F, _F, _F, _F, _I, _I, _I, _I, _I, _I,
                               _P, _P, _P, _P, _I, _P, _Z, _I, _I, _P]),
    "hvpr_pillar_vfe_fwd_f32": (_I, [_P, _P, _P, _I, _I, _P, _F, _F, _F, _F, _F, _F, _P, _P, _P, _P, _P, _P, _P, _P,
                                     _P, _P, _P, _P]),
    "hvpr_memory_bank_packed_floats": (_Z, [_I]),
    "hvpr_memory_bank_pack_f32": (_I, [_P, _I, _P, _P]),
    "hvpr_memory_readout_fwd_f32": (_I, [_P, _I, _P, _P, _P, _I, _I, _P, _P, _P]),
    "hvpr_scatter_workspace_bytes": (_Z, [_I, _I, _I]),
    "hvpr_memory_scatter_fwd_f32": (_I, [_P, _P, _P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _Z, _P]),
    "hvpr_encode_fwd_f32": (_I, [_P, _I, _I, _I, _I, _P, _I, _F, _F, _F, _F, _F, _F, _I, _I, _I, _I, _I, _I, _F, _F, _F,
                                 _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P,
                                 _P, _P, _Z, _I, _I, _I, _P]),
    "hvpr_scatter_bev_fwd_f32": (_I, [_P, _I, _P, _I, _P, _I, _P, _I, _P, _I, _I, _I, _P, _P, _P, _Z, _P]),
    "hvpr_spatial_gate_f32": (_I, [_P, _I, _I, _I, _I, _P, _F, _F, _F, _P, _P]),
    "hvpr_head_decode_f32": (_I, [_P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _F, _F, _F, _P, _P, _P, _P, _P]),
    "hvpr_score_topk_workspace_bytes": (_Z, [_I, _I]),
    "hvpr_score_topk_f32": (_I, [_P, _I, _I, _F, _I, _I, _P, _P, _P, _P, _Z, _P]),
    "hvpr_nms_workspace_bytes": (_Z, [_I]),
    "hvpr_nms_bev_f32": (_I, [_P, _I, _P, _P, _I, _F, _I, _I, _P, _P, _P, _Z, _P]),
    "hvpr_gather_predictions_f32": (_I, [_P, _I, _P, _P, _P, _I, _P, _P, _P, _P, _P]),
    "hvpr_boxes_pairwise_f32": (_I, [_P, _I, _P, _I, _I, _P, _P]),
    "hvpr_furthest_point_sample_f32": (_I, [_P, _I, _I, _I, _P, _P]),
    "hvpr_ball_query_f32": (_I, [_P, _P, _I, _I, _I, _F, _I, _P, _P]),
    "hvpr_three_nn_f32": (_I, [_P, _P, _I, _I, _I, _P, _P, _P]),
    "hvpr_group_points_f32": (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _P]),
    "hvpr_group_points_grad_f32": (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _P]),
    "hvpr_three_interpolate_f32": (_I, [_P, _P, _P, _I, _I, _I, _I, _P, _P]),
    "hvpr_three_interpolate_grad_f32": (_I, [_P, _P, _P, _I, _I, _I, _I, _P, _P]),
    "hvpr_group_rows_f32": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P]),
    "hvpr_group_rows_grad_f32": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P, _P]),
    "hvpr_max_samples_f32": (_I, [_P, _c.c_longlong, _I, _I, _P, _P, _P]),
    "hvpr_max_samples_grad_f32": (_I, [_P, _P, _c.c_longlong, _I, _I, _P, _P]),
    "hvpr_fp_rows_f32": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P]),
    "hvpr_fp_rows_grad_f32": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P]),
    "hvpr_spatial_gate_train_workspace_bytes": (_Z, [_I, _I, _I]),
    "hvpr_spatial_gate_train_fwd_f32": (_I, [_P, _I, _I, _I, _I, _P, _P, _P, _P, _F, _P, _P, _P, _P, _P, _P, _Z, _P]),
    "hvpr_spatial_gate_train_bwd_f32": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _Z, _P]),
    "hvpr_conv2d_wgrad_workspace_bytes": (_Z, [_I, _I, _I, _I, _I, _I, _I]),
    "hvpr_conv2d_wgrad_nhwc_f32": (_I, [_P, _I, _I, _I, _I, _P, _I, _I, _I, _P, _P, _Z, _P]),
    "hvpr_bn_workspace_bytes": (_Z, [_c.c_longlong, _I]),
    "hvpr_bn_stats_nhwc_f32": (_I, [_P, _c.c_longlong, _I, _F, _P, _P, _P, _P, _Z, _P]),
    "hvpr_bn_relu_fwd_nhwc_f32": (_I, [_P, _c.c_longlong, _I, _P, _P, _I, _P, _P, _P, _P]),
    "hvpr_bn_relu_bwd_nhwc_f32": (_I, [_P, _P, _c.c_longlong, _I, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _Z, _P]),
    "hvpr_bn_relu_bwd_sums_nhwc_f32": (_I, [_P, _P, _c.c_longlong, _I, _P, _P, _P, _P, _I, _P, _P, _P, _P, _Z, _P]),
    "hvpr_bn_relu_bwd_apply_nhwc_f32": (_I, [_P, _P, _c.c_longlong, _I, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _c.c_double, _P]),
    "hvpr_memory_train_workspace_bytes": (_Z, [_I]),
    "hvpr_memory_train_fwd_f32": (_I, [_P, _c.c_longlong, _P, _I, _F, _P, _P, _P, _Z, _P]),
    "hvpr_memory_train_bwd_f32": (_I, [_P, _P, _c.c_longlong, _P, _I, _F, _P, _P, _P, _P, _P, _Z, _P]),
    "hvpr_point_pillar_topk_f32": (_I, [_P, _I, _P, _P, _I, _I, _P, _P]),
    "hvpr_scatter_add_rows_f32": (_I, [_P, _P, _c.c_longlong, _I, _I, _P, _P]),
    "hvpr_segment_sum_rows_f32": (_I, [_P, _c.c_longlong, _I, _I, _P, _P, _P, _c.c_longlong, _P, _c.c_longlong, _P]),
    "hvpr_fused_adam_truewd_f32": (_I, [_P, _P, _P, _P, _c.c_longlong, _F, _F, _F, _F, _F, _I, _P, _P]),
    "hvpr_assign_targets_workspace_bytes": (_Z, [_I, _I]),
    "hvpr_assign_targets_f32": (_I, [_P, _I, _P, _I, _I, _I, _I, _F, _F, _I, _I, _I, _c.c_longlong, _P, _P, _P, _P, _P, _Z, _P]),
    "hvpr_rpn_losses_workspace_bytes": (_Z, [_I, _c.c_longlong]),
    "hvpr_rpn_losses_f32": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _c.c_longlong, _I, _I, _F, _F, _F, _P, _F, _F, _F, _F, _P, _P, _P, _P,
                                 _P, _Z, _P]),
    "hvpr_mse_loss_workspace_bytes": (_Z, []),
    "hvpr_mse_loss_f32": (_I, [_P, _P, _c.c_longlong, _I, _F, _P, _P, _P, _Z, _P]),
    "hvpr_split_bf16_f32": (_I, [_P, _c.c_longlong, _I, _P, _P]),
    "hvpr_unsplit_bf16_f32": (_I, [_P, _c.c_longlong, _I, _P, _P]),
    "hvpr_conv2d_nhwc_bf16x3": (_I, [_P, _I, _I, _I, _I, _P, _P, _I, _I, _I, _I, _P, _P, _I, _P, _I, _I, _I, _I, _I, _P]),
    "hvpr_deconv_nhwc_bf16x3": (_I, [_P, _I, _I, _I, _I, _P, _P, _I, _I, _I, _I, _P, _I, _I, _I, _P]),
    "hvpr_point_flags_f32": (_I, [_P, _I, _I, _I, _P, _F, _P, _P, _I, _I, _P, _P]),
    "hvpr_compact_workspace_bytes": (_Z, [_I]),
    "hvpr_compact_rows_f32": (_I, [_P, _I, _I, _P, _P, _I, _P, _P, _Z, _P]),
    "hvpr_frame_offsets_f32": (_I, [_P, _I, _I, _I, _P, _P]),
    "hvpr_gather_rows_f32": (_I, [_P, _I, _I, _P, _I, _P, _P]),
    "hvpr_pillar_vfe_train_workspace_bytes": (_Z, []),
    "hvpr_pillar_vfe_train_fwd_f32": (_I, [_P, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _F, _F, _F, _F, _F, _F, _F, _P, _P, _P, _P, _P, _P, _Z, _P]),
    "hvpr_pillar_vfe_bwd_f32": (_I, [_P, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _F, _F, _F, _F, _F, _F, _F, _P, _P, _P, _P, _P, _P, _P, _P, _Z, _P]),
    "hvpr_conv2d_wino_wgrad_workspace_bytes": (_Z, [_I, _I, _I, _I, _I]),
    "hvpr_conv2d_wino_wgrad_nhwc_f32": (_I, [_P, _I, _I, _I, _I, _P, _I, _P, _P, _Z, _P]),
    "hvpr_conv2d_wino_packed_floats": (_Z, [_I, _I]),
    "hvpr_conv2d_wino_pack_f32": (_I, [_P, _P, _I, _I, _I, _P, _P]),
    "hvpr_conv2d_wino_nhwc_f32": (_I, [_P, _I, _I, _I, _I, _P, _P, _I, _I, _P, _P, _I, _P, _I, _I, _I, _P, _P]),
    "hvpr_conv2d_wino_stats_rows": (_I, [_I, _I, _I]),
    "hvpr_conv2d_s2_dgrad_nhwc_f32": (_I, [_P, _I, _I, _I, _I, _P, _P, _I, _I, _I, _I, _P, _I, _I, _P]),
    "hvpr_bn_relu_fwd_slice_nhwc_f32": (_I, [_P, _c.c_longlong, _I, _P, _P, _I, _P, _I, _I, _P]),
    "hvpr_bn_relu_bwd_slice_nhwc_f32": (_I, [_P, _I, _I, _P, _c.c_longlong, _I, _P, _P, _P, _P, _I, _P, _P, _P, _P, _Z, _P]),
    "hvpr_bn_train_affine_f32": (_I, [_P, _P, _P, _I, _P, _P, _F, _F, _P, _P, _P, _P, _P, _P]),
    "hvpr_bn_finalize_partials_f32": (_I, [_P, _I, _I, _c.c_longlong, _F, _P, _P, _P, _P]),
    "hvpr_conv2d_nhwc_f32": (_I, [_P, _I, _I, _I, _I, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _I, _P, _I, _I, _I, _P]),
}

_lib = None


class HvprLibraryError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HvprLibraryError(
                f"{LIB_PATH} is missing: build it with `python -m hvpr_amd.build` (hipcc --offload-arch=gfx950). "
                "hvpr_amd has no CPU fallback.")
        try:
            L = ctypes.CDLL(LIB_PATH)
        except OSError as e:  # pragma: no cover
            raise HvprLibraryError(f"cannot load {LIB_PATH}: {e}") from e
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        if L.hvpr_abi_version() != ABI_VERSION:      # a stale .so next to newer Python wrappers (buffer sizes, argument lists)
            raise HvprLibraryError(f"{LIB_PATH} has ABI version {L.hvpr_abi_version()}, these wrappers need {ABI_VERSION}: "
                                   "rebuild it with `python -m hvpr_amd.build`")
        _lib = L
    return _lib


def check(status, what):
    if status != 0:
        msg = lib().hvpr_status_string(int(status))
        raise RuntimeError(f"{what} failed: {msg.decode() if msg else status}")
