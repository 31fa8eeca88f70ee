"""ctypes mirror of include/dynamo_hip.h (the C ABI of libdynamo_hip.so).

Only plain pointers / ints / floats cross this boundary -- no torch types.  Tensors are handed over
as `tensor.data_ptr()`; the caller (hipops.functions) validates dtype / contiguity / device first.
"""
import ctypes as C

DD_MAX_SCALES = 4
DD_NUM_SRC = 2
DD_ABI_VERSION = 2
DD_MODE_RIGID, DD_MODE_FLOW, DD_MODE_FLOW_MASK = 0, 1, 2
DD_PARTIAL_STRIDE = 40
DD_SUMS_STRIDE = 8

_fp = C.c_void_p        # device (or, for the host-math test library, host) float*


class DDPhotoScale(C.Structure):
    _fields_ = [
        ("shift", C.c_int), ("h", C.c_int), ("w", C.c_int),
        ("w_photo", C.c_float), ("w_cons", C.c_float),
        ("disp", _fp),
        ("flow", _fp * DD_NUM_SRC),
        ("mask", _fp * DD_NUM_SRC),
        ("noise", _fp),
        ("g_disp", _fp),
        ("g_flow", _fp * DD_NUM_SRC),
        ("g_mask", _fp * DD_NUM_SRC),
        ("out_color", _fp * DD_NUM_SRC),
        ("out_sample", _fp * DD_NUM_SRC),
        ("out_depth", _fp),
        ("out_idsel", _fp),
        ("out_resid", _fp * DD_NUM_SRC),
        ("out_delta", _fp * DD_NUM_SRC),
    ]


class DDPhotoArgs(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int),
        ("B", C.c_int), ("H", C.c_int), ("W", C.c_int),
        ("num_scales", C.c_int), ("mode", C.c_int), ("automask", C.c_int), ("want_grad", C.c_int),
        ("min_depth", C.c_float), ("max_depth", C.c_float), ("ssim_weight", C.c_float),
        ("eps", C.c_float), ("disp_thr", C.c_float),
        ("target", _fp),
        ("source", _fp * DD_NUM_SRC),
        ("source_packed", _fp * DD_NUM_SRC),
        ("K", _fp), ("inv_K", _fp),
        ("T", _fp * DD_NUM_SRC),
        ("ts", _fp * DD_NUM_SRC),
        ("g_T", _fp * DD_NUM_SRC),
        ("sums", _fp),
        ("workspace", _fp),
        ("scale", DDPhotoScale * DD_MAX_SCALES),
    ]


DD_NUM_TERMS = 7
DD_MAX_RES = 128
TERM_NAMES = ("p_photo", "d_smooth", "d_ground", "c_smooth", "c_consistency", "m_sparsity", "m_smooth")   # options.py g_* order


class DDAssembleArgs(C.Structure):
    _fields_ = [
        ("n", C.c_int), ("num_scales", C.c_int),
        ("coef", C.c_float * DD_NUM_TERMS),
        ("norm", C.c_float * DD_MAX_RES),
        ("term_of", C.c_int8 * DD_MAX_RES),
        ("scale_of", C.c_int8 * DD_MAX_RES),
    ]


DD_REG_SMOOTH = 5
DD_REG_RES_STRIDE = 16


class DDRegSmooth(C.Structure):
    _fields_ = [("inp", _fp), ("g_inp", _fp), ("C", C.c_int), ("normalise", C.c_int), ("weight", C.c_float)]


class DDRegScale(C.Structure):
    _fields_ = [
        ("h", C.c_int), ("w", C.c_int),
        ("img", _fp),
        ("smooth", DDRegSmooth * DD_REG_SMOOTH),
        ("delta", _fp * DD_NUM_SRC), ("delta_sum", _fp * DD_NUM_SRC), ("prob", _fp * DD_NUM_SRC), ("g_prob", _fp * DD_NUM_SRC),
        ("w_sparsity", C.c_float * DD_NUM_SRC),
        ("disp", _fp), ("g_disp", _fp), ("inv_K", _fp), ("rand_idx", _fp), ("plane", _fp),
        ("w_ground", C.c_float),
    ]


class DDRegArgs(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int), ("B", C.c_int), ("num_scales", C.c_int),
        ("np_per_it", C.c_int), ("max_it", C.c_int),
        ("tol", C.c_float), ("g_prior", C.c_float), ("min_depth", C.c_float), ("max_depth", C.c_float),
        ("res", _fp), ("workspace", _fp),
        ("scale", DDRegScale * DD_MAX_SCALES),
    ]


class DDJpegHeader(C.Structure):
    """include/dynamo_hip.h DDJpegHeader."""
    _fields_ = [("data_offset", C.c_int32), ("data_end", C.c_int32), ("width", C.c_int32), ("height", C.c_int32), ("restart_interval", C.c_int32),
                ("ncomp", C.c_int32), ("h", C.c_int32 * 3), ("v", C.c_int32 * 3), ("tq", C.c_int32 * 3), ("td", C.c_int32 * 3), ("ta", C.c_int32 * 3),
                ("reserved", C.c_int32 * 3), ("qt", (C.c_uint16 * 64) * 4), ("bits", (C.c_uint8 * 16) * 4), ("vals", (C.c_uint8 * 256) * 4)]


def ptr(t):
    """Raw address of a tensor's storage (None -> NULL)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def declare(lib):
    """Attach argtypes/restypes for every entry point of include/dynamo_hip.h."""
    i, f, v, z = C.c_int, C.c_float, C.c_void_p, C.c_size_t
    sig = {
        "dd_photo_loss": (i, [C.POINTER(DDPhotoArgs), v]),
        "dd_photo_workspace_bytes": (z, [C.POINTER(DDPhotoArgs)]),
        "dd_photo_timing": (i, [i]),
        "dd_photo_timing_read": (i, [C.POINTER(C.c_float), C.POINTER(i), i]),
        "dd_smooth_loss": (i, [v, v, i, i, i, i, i, f, v, v, v, v]),
        "dd_smooth_workspace_bytes": (z, [i, i, i, i]),
        "dd_sparsity_loss": (i, [v, v, v, i, i, i, f, v, v, v, v]),
        "dd_sparsity_workspace_bytes": (z, [i, i, i]),
        "dd_ground_loss": (i, [v, v, v, i, i, i, i, i, f, f, f, f, f, v, v, v, v, v]),
        "dd_ground_workspace_bytes": (z, [i, i, i, i]),
        "dd_resize_workspace_bytes": (z, [i, i, i, i, i]),
        "dd_resize_bicubic": (i, [v, i, i, i, v, v, i, i, v, v, i, v, v, i, v, z, v]),
        "dd_ground_candidates": (i, [v, v, v, i, i, i, i, i, f, f, f, v, v]),
        "dd_ground_select": (i, [v, v, v, i, i, i, i, f, f, f, f, f, v, v, v, v, v, v]),
        "dd_ground_plane": (i, [v, v, i, i, i, i, i, f, f, v, v, v, v]),
        "dd_assemble_losses": (i, [v, C.POINTER(DDAssembleArgs), v, v, v]),
        "dd_reg_losses": (i, [C.POINTER(DDRegArgs), v]),
        "dd_reg_losses_finish": (i, [C.POINTER(DDRegArgs), C.POINTER(DDAssembleArgs), v, v, v]),
        "dd_fused_loss": (i, [C.POINTER(DDPhotoArgs), C.POINTER(DDRegArgs), C.POINTER(DDAssembleArgs), v, v, v]),
        "dd_fused_loss_part": (i, [C.POINTER(DDPhotoArgs), C.POINTER(DDRegArgs), C.POINTER(DDAssembleArgs), v, v, v, i]),
        "dd_fused_loss_supported": (i, [C.POINTER(DDPhotoArgs), C.POINTER(DDRegArgs)]),
        "dd_reg_workspace_bytes": (z, [C.POINTER(DDRegArgs)]),
        "dd_jpeg_workspace_bytes": (z, [i, i, i, i, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
        "dd_jpeg_decode": (i, [v, C.c_longlong, v, i, i, i, i, C.POINTER(C.c_int), C.POINTER(C.c_int), v, v, z, v]),
        "dd_backproject": (i, [v, v, i, i, i, v, v]),
        "dd_backproject_bwd": (i, [v, v, i, i, i, v, v]),
        "dd_project3d": (i, [v, v, v, i, i, i, f, v, v, v]),
        "dd_project3d_bwd": (i, [v, v, v, v, v, i, i, i, f, v, v, v, v]),
        "dd_project3d_workspace_bytes": (z, [i, i, i]),
        "dd_ssim": (i, [v, v, i, i, i, i, v, v]),
        "dd_ssim_bwd": (i, [v, v, v, i, i, i, i, v, v, v]),
        "dd_disp_to_depth": (i, [v, z, f, f, v, v, v]),
        "dd_pose_matrix": (i, [v, v, i, i, v, v]),
        "dd_pose_matrix_bwd": (i, [v, v, v, i, i, v, v, v]),
        "dd_channel_sum_nhwc": (i, [v, C.c_longlong, i, v, v, v]),
        "dd_channel_sum_workspace_bytes": (z, [i]),
        "dd_reflect_pad1_nhwc": (i, [v, i, i, i, i, v, v]),
        "dd_reflect_pad1_nhwc_bwd": (i, [v, i, i, i, i, v, v]),
        "dd_dwconv3x3_nhwc": (i, [v, v, i, i, i, i, i, v, v]),
        "dd_dwconv3x3_nhwc_bwd_data": (i, [v, v, i, i, i, i, i, v, v]),
        "dd_dwconv3x3_nhwc_bwd_weight": (i, [v, v, i, i, i, i, i, v, v, z, v]),
        "dd_dwconv3x3_workspace_bytes": (z, [i, i, i]),
        "dd_conv3x3_cout1_bwd_data": (i, [v, v, i, i, i, i, i, v, v]),
        "dd_prepare_frames": (i, [v, v, v, i, i, i, i, v, v, v, v]),
        "dd_prepare_frames_workspace_bytes": (z, [i, i]),
        "dd_pyramid_down2": (i, [v, i, i, i, v, v]),
        "dd_pack_rgb": (i, [v, i, i, i, v, v]),
        "dd_depth_metrics": (i, [v, i, i, i, v, v, i, v, C.POINTER(C.c_double), f, f, v, v, v, z, v]),
        "dd_depth_metrics_workspace_bytes": (z, [i, i]),
        "dd_depth_metrics_masked": (i, [v, i, i, i, v, v, i, v, C.POINTER(C.c_double), f, f, v, i, i, v, v, v, v, z, v]),
        "dd_depth_metrics_masked_workspace_bytes": (z, [i, i]),
        "dd_bn_act_fwd": (i, [v, v, C.c_longlong, i, v, v, f, f, v, v, v, v, i, v, v, z, v]),
        "dd_bn_act_bwd": (i, [v, v, v, C.c_longlong, i, v, v, v, v, i, v, v, v, v, v, z, v]),
        "dd_bn_workspace_bytes": (z, [i]),
        "dd_bn_act_fwd_t": (i, [v, v, C.c_longlong, i, v, v, f, f, v, v, v, v, i, v, i, v, z, v]),
        "dd_bn_act_bwd_t": (i, [v, v, v, C.c_longlong, i, v, v, v, v, i, v, v, v, v, i, v, z, v]),
        "dd_channel_sum_nhwc_t": (i, [v, C.c_longlong, i, v, i, v, v]),
        "dd_reflect_pad1_nhwc_t": (i, [v, i, i, i, i, v, i, v]),
        "dd_reflect_pad1_nhwc_bwd_t": (i, [v, i, i, i, i, v, i, v]),
        "dd_up_cat_pad_t": (i, [v, v, i, i, i, i, i, i, i, v, i, v]),
        "dd_up_cat_pad_bwd_t": (i, [v, v, i, i, i, i, i, i, i, v, v, i, v]),
        "dd_layer_norm_fwd": (i, [v, C.c_longlong, i, v, v, f, v, v, v, v]),
        "dd_layer_norm_bwd": (i, [v, v, v, v, v, C.c_longlong, i, v, v, v, z, v]),
        "dd_layer_norm_workspace_bytes": (z, [i]),
        "dd_layer_scale_bwd": (i, [v, v, v, i, i, i, v, v, v, z, v]),
        "dd_layer_scale_workspace_bytes": (z, [i, i]),
        "dd_layer_norm_fwd_t": (i, [v, C.c_longlong, i, v, v, f, v, v, v, i, v]),
        "dd_layer_norm_bwd_t": (i, [v, v, v, v, v, C.c_longlong, i, v, v, v, z, i, v]),
        "dd_layer_scale_bwd_t": (i, [v, v, v, i, i, i, v, v, v, z, i, v]),
        "dd_layer_scale_fwd_t": (i, [v, v, v, i, i, i, v, i, v]),
        "dd_dwconv3x3_nhwc_t": (i, [v, v, i, i, i, i, i, v, i, v]),
        "dd_dwconv3x3_nhwc_bwd_data_t": (i, [v, v, i, i, i, i, i, v, i, v]),
        "dd_dwconv3x3_nhwc_bwd_weight_t": (i, [v, v, i, i, i, i, i, v, v, z, i, v]),
        "dd_redu_supported": (i, [i, i]),
        "dd_redu_workspace_bytes": (z, [C.c_longlong, i, i]),
        "dd_redu_fwd": (i, [v, v, v, v, C.c_longlong, i, i, v, v]),
        "dd_redu_bwd_data": (i, [v, v, C.c_longlong, i, i, v, v, v]),
        "dd_redu_bwd_weight": (i, [v, v, v, C.c_longlong, i, i, v, v, v, z, v]),
        "dd_conv_head_supported": (i, [i]),
        "dd_conv_head_workspace_bytes": (z, [i, i, i, i]),
        "dd_conv_head_fwd": (i, [v, v, C.c_longlong, C.c_longlong, C.c_longlong, v, i, i, i, i, v, v]),
        "dd_conv_head_bwd_weight": (i, [v, v, i, i, i, i, v, v, v, z, v]),
        "dd_conv_small_supported": (i, [i, i, i]),
        "dd_conv_small_workspace_bytes": (z, [i, i, i]),
        "dd_conv_small_fwd": (i, [v, v, C.c_longlong, C.c_longlong, C.c_longlong, C.c_longlong, v, i, i, i, i, i, i, v, v, z, v]),
        "dd_conv_small_bwd_data": (i, [v, v, C.c_longlong, C.c_longlong, C.c_longlong, C.c_longlong, i, i, i, i, i, i, v, v, z, v]),
        "dd_conv_small_bwd_weight": (i, [v, v, i, i, i, i, i, i, v, v, v, z, v]),
        "dd_conv3x3_mfma_supported": (i, [i, i]),
        "dd_conv3x3_mfma_pack_bytes": (z, [i, i]),
        "dd_conv3x3_mfma_pack": (i, [v, C.c_longlong, C.c_longlong, C.c_longlong, C.c_longlong, i, i, v, v, v]),
        "dd_conv3x3_mfma_pack_many_job_words": (i, []),
        "dd_conv3x3_mfma_pack_many_blocks": (i, [i, i, i, i]),
        "dd_conv3x3_mfma_pack_many": (i, [v, v, i, v]),
        "dd_conv3x3_mfma": (i, [v, v, v, i, i, i, i, i, i, v, v]),
        "dd_conv3x3_mfma_n": (i, [v, v, v, i, i, i, i, i, i, i, v, v]),
        "dd_conv3x3_mfma_flat_supported": (i, [i, i, i, i, i]),
        "dd_conv3x3_mfma_flat_workspace_bytes": (z, [i, i, i, i, i]),
        "dd_conv3x3_mfma_flat": (i, [v, v, v, i, i, i, i, i, v, v, z, v]),
        "dd_conv3x3_mfma_flat_n": (i, [v, v, v, i, i, i, i, i, i, v, v, z, v]),
        "dd_conv3x3_mfma_wgrad_workspace_bytes": (z, [i, i, i, i, i]),
        "dd_conv3x3_mfma_bwd_weight": (i, [v, v, i, i, i, i, i, i, v, v, z, v]),
        "dd_conv3x3_mfma_bwd_weight_n": (i, [v, v, i, i, i, i, i, i, i, v, v, z, v]),
        "dd_conv3x3_half_supported": (i, [i, i]),
        "dd_conv3x3_half_pack_bytes": (z, [i, i]),
        "dd_conv3x3_half_pack": (i, [v, C.c_longlong, C.c_longlong, C.c_longlong, C.c_longlong, i, i, i, v, v, v]),
        "dd_conv3x3_half": (i, [v, v, v, i, i, i, i, i, i, i, v, v]),
        "dd_conv3x3_half_wgrad_workspace_bytes": (z, [i, i, i, i, i]),
        "dd_conv3x3_half_bwd_weight": (i, [v, v, i, i, i, i, i, i, i, v, v, z, v]),
        "dd_pw_gemm_pack_bytes": (z, [i, i]),
        "dd_mlp_pack": (i, [v, C.c_longlong, C.c_longlong, v, C.c_longlong, C.c_longlong, i, i, v, v, v, v, v, v]),
        "dd_mlp_fwd_supported": (i, [i]),
        "dd_mlp_fwd": (i, [v, v, v, v, v, i, i, v, v]),
        "dd_pw_gemm": (i, [v, v, v, i, i, i, i, v, v]),
        "dd_gelu_pair": (i, [v, v, v, z, v]),
        "dd_adam_chunk": (i, []),
        "dd_adam_multi": (i, [v, i, v, i, v, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, v, v, v]),
        "dd_error_string": (C.c_char_p, [i]),
        "dd_abi_version": (i, []),
    }
    found = []
    for name, (res, args) in sig.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            continue
        fn.restype = res
        fn.argtypes = args
        found.append(name)
    return found


EXPORTED = (
    "dd_photo_loss", "dd_photo_workspace_bytes", "dd_photo_timing", "dd_photo_timing_read", "dd_photo_loss_part", "dd_smooth_loss", "dd_smooth_workspace_bytes",
    "dd_sparsity_loss", "dd_sparsity_workspace_bytes", "dd_ground_loss", "dd_ground_workspace_bytes", "dd_ground_plane", "dd_ground_candidates", "dd_ground_select",
    "dd_assemble_losses", "dd_reg_losses", "dd_reg_losses_finish", "dd_fused_loss", "dd_fused_loss_part", "dd_fused_loss_supported", "dd_reg_workspace_bytes", "dd_backproject", "dd_backproject_bwd", "dd_project3d", "dd_project3d_bwd", "dd_project3d_workspace_bytes",
    "dd_ssim", "dd_ssim_bwd", "dd_disp_to_depth", "dd_pose_matrix", "dd_pose_matrix_bwd",
    "dd_channel_sum_nhwc", "dd_channel_sum_workspace_bytes", "dd_reflect_pad1_nhwc", "dd_reflect_pad1_nhwc_bwd",
    "dd_dwconv3x3_nhwc", "dd_dwconv3x3_nhwc_bwd_data", "dd_dwconv3x3_nhwc_bwd_weight", "dd_dwconv3x3_workspace_bytes", "dd_conv3x3_cout1_bwd_data",
    "dd_prepare_frames", "dd_prepare_frames_workspace_bytes", "dd_pyramid_down2", "dd_pack_rgb", "dd_depth_metrics", "dd_depth_metrics_workspace_bytes", "dd_depth_metrics_masked", "dd_depth_metrics_masked_workspace_bytes", "dd_bn_act_fwd", "dd_bn_act_bwd", "dd_bn_workspace_bytes",
    "dd_bn_act_fwd_t", "dd_bn_act_bwd_t", "dd_channel_sum_nhwc_t", "dd_reflect_pad1_nhwc_t", "dd_reflect_pad1_nhwc_bwd_t", "dd_up_cat_pad_t", "dd_up_cat_pad_bwd_t",
    "dd_layer_norm_fwd", "dd_layer_norm_bwd", "dd_layer_norm_workspace_bytes", "dd_layer_scale_bwd", "dd_layer_scale_workspace_bytes",
    "dd_jpeg_workspace_bytes", "dd_jpeg_decode", "dd_resize_workspace_bytes", "dd_resize_bicubic", "dd_layer_norm_fwd_t", "dd_layer_norm_bwd_t", "dd_layer_scale_bwd_t", "dd_layer_scale_fwd_t", "dd_dwconv3x3_nhwc_t", "dd_dwconv3x3_nhwc_bwd_data_t",
    "dd_dwconv3x3_nhwc_bwd_weight_t", "dd_adam_chunk", "dd_adam_multi", "dd_conv_small_supported", "dd_conv_small_workspace_bytes",
    "dd_conv_small_fwd", "dd_conv_small_bwd_data", "dd_conv_small_bwd_weight", "dd_conv_head_supported", "dd_conv_head_workspace_bytes", "dd_conv_head_fwd",
    "dd_conv_head_bwd_weight", "dd_redu_supported", "dd_redu_workspace_bytes", "dd_redu_fwd", "dd_redu_bwd_data", "dd_redu_bwd_weight",
    "dd_conv3x3_mfma_supported", "dd_conv3x3_mfma_pack_bytes", "dd_conv3x3_mfma_pack", "dd_conv3x3_mfma_pack_many_job_words",
    "dd_conv3x3_mfma_pack_many_blocks", "dd_conv3x3_mfma_pack_many", "dd_conv3x3_mfma", "dd_conv3x3_mfma_n",
    "dd_conv3x3_mfma_flat_supported", "dd_conv3x3_mfma_flat_workspace_bytes", "dd_conv3x3_mfma_flat", "dd_conv3x3_mfma_flat_n",
    "dd_conv3x3_mfma_wgrad_workspace_bytes", "dd_conv3x3_mfma_bwd_weight", "dd_conv3x3_mfma_bwd_weight_n",
    "dd_conv3x3_half_supported", "dd_conv3x3_half_pack_bytes", "dd_conv3x3_half_pack", "dd_conv3x3_half",
    "dd_conv3x3_half_wgrad_workspace_bytes", "dd_conv3x3_half_bwd_weight",
    "dd_pw_gemm_pack_bytes", "dd_mlp_pack", "dd_pw_gemm", "dd_gelu_pair", "dd_mlp_fwd_supported", "dd_mlp_fwd",
    "dd_error_string", "dd_abi_version",
)


def fill_photo_args(*, B, H, W, mode, automask, want_grad, min_depth, max_depth, ssim_weight, eps, disp_thr,
                    target, source, K, inv_K, T, ts, g_T, sums, workspace, scales, source_packed=None):
    """Builds a DDPhotoArgs from tensors.  `scales` is a list of dicts with the DDPhotoScale field names
    (tensors or None; per-frame entries as 2-lists)."""
    a = DDPhotoArgs()
    a.abi_version = DD_ABI_VERSION
    a.B, a.H, a.W = B, H, W
    a.num_scales = len(scales)
    a.mode, a.automask, a.want_grad = mode, int(bool(automask)), int(bool(want_grad))
    a.min_depth, a.max_depth, a.ssim_weight, a.eps, a.disp_thr = min_depth, max_depth, ssim_weight, eps, disp_thr
    a.target = ptr(target)
    a.K, a.inv_K = ptr(K), ptr(inv_K)
    a.sums, a.workspace = ptr(sums), ptr(workspace)
    for f in range(DD_NUM_SRC):
        a.source[f] = ptr(source[f])
        a.source_packed[f] = ptr(source_packed[f]) if source_packed is not None else None
        a.T[f] = ptr(T[f])
        a.ts[f] = ptr(ts[f]) if ts is not None else None
        a.g_T[f] = ptr(g_T[f]) if g_T is not None else None
    for s, d in enumerate(scales):
        sc = a.scale[s]
        sc.shift, sc.h, sc.w = d["shift"], d["h"], d["w"]
        sc.w_photo, sc.w_cons = d.get("w_photo", 0.0), d.get("w_cons", 0.0)
        for name in ("disp", "noise", "g_disp", "out_depth", "out_idsel"):
            setattr(sc, name, ptr(d.get(name)))
        for name in ("flow", "mask", "g_flow", "g_mask", "out_color", "out_sample", "out_resid", "out_delta"):
            pair = d.get(name) or (None, None)
            arr = getattr(sc, name)
            for f in range(DD_NUM_SRC):
                arr[f] = ptr(pair[f])
    return a
