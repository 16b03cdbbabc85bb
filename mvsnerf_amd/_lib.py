"""ctypes binding of libmvsnerf_hip.so (the C ABI declared in include/mvsnerf_hip.h).

The product path has NO fallback: if the shared library is missing or a call fails, a RuntimeError
naming the op is raised.  PyTorch is used only for device memory and streams.
"""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libmvsnerf_hip.so")
CSRC = os.path.join(_HERE, "csrc")

_c_fp = ctypes.c_void_p      # device pointers travel as integers
_c_i = ctypes.c_int
_c_l = ctypes.c_int64


class RaymarchArgs(ctypes.Structure):
    """mvsnerf_raymarch_args (include/mvsnerf_hip.h)."""
    _fields_ = [
        ("vol", _c_fp), ("D", _c_i), ("H", _c_i), ("W", _c_i),
        ("imgs", _c_fp), ("V", _c_i), ("IH", _c_i), ("IW", _c_i),
        ("w2c", _c_fp), ("K", _c_fp), ("packed_mlp", _c_fp),
        ("rays_pts", _c_fp), ("rays_ndc", _c_fp), ("z_vals", _c_fp), ("rays_dir", _c_fp),
        ("N", _c_l), ("S", _c_i), ("white_bkgd", _c_i),
        ("dirs_tmp", _c_fp), ("input_feat", _c_fp), ("raw", _c_fp),
        ("rgb_map", _c_fp), ("disp", _c_fp), ("acc", _c_fp), ("weights", _c_fp), ("depth", _c_fp), ("alpha", _c_fp),
        ("packed_mlp_bf16", _c_fp), ("imgs_nhwc4", _c_fp), ("packed_mlp_split", _c_fp), ("n_split", _c_i), ("guard", _c_fp), ("vol_layout", _c_i),
    ]


class RaymarchTrainArgs(ctypes.Structure):
    """mvsnerf_raymarch_train_args (include/mvsnerf_hip.h)."""
    _fields_ = [
        ("vol", _c_fp), ("D", _c_i), ("H", _c_i), ("W", _c_i), ("C", _c_i),
        ("imgs_nhwc4", _c_fp), ("V", _c_i), ("IH", _c_i), ("IW", _c_i),
        ("w2c", _c_fp), ("K", _c_fp), ("packed_mlp", _c_fp), ("packed_mlp_bf16", _c_fp), ("bf16", _c_i),
        ("rays_pts", _c_fp), ("rays_ndc", _c_fp), ("z_vals", _c_fp), ("rays_dir", _c_fp),
        ("N", _c_l), ("S", _c_i), ("white_bkgd", _c_i),
        ("dirs_tmp", _c_fp), ("input_feat", _c_fp), ("raw", _c_fp), ("saved", _c_fp),
        ("rgb_map", _c_fp), ("disp", _c_fp), ("acc", _c_fp), ("weights", _c_fp), ("depth", _c_fp), ("alpha", _c_fp), ("vol_layout", _c_i),
    ]


class RaymarchBwdArgs(ctypes.Structure):
    """mvsnerf_raymarch_bwd_args (include/mvsnerf_hip.h)."""
    _fields_ = [
        ("packed_mlp", _c_fp), ("packed_bwd", _c_fp), ("bf16", _c_i), ("F", _c_i),
        ("raw", _c_fp), ("saved", _c_fp), ("z_vals", _c_fp), ("rays_ndc", _c_fp),
        ("N", _c_l), ("S", _c_i), ("white_bkgd", _c_i),
        ("g_rgb", _c_fp), ("g_depth", _c_fp), ("g_weights", _c_fp), ("g_alpha", _c_fp),
        ("d_raw", _c_fp), ("gslots", _c_fp), ("d_feat", _c_fp), ("n_feat_out", _c_i),
        ("gw", ctypes.POINTER(_c_fp)), ("gb", ctypes.POINTER(_c_fp)), ("maps", _c_fp), ("workspace", _c_fp),
        ("gvol", _c_fp), ("D", _c_i), ("H", _c_i), ("W", _c_i), ("C", _c_i),
    ]


class RenderArgs(ctypes.Structure):
    """mvsnerf_render_args (include/mvsnerf_hip.h)."""
    _fields_ = [
        ("vol", _c_fp), ("D", _c_i), ("H", _c_i), ("W", _c_i),
        ("imgs_nhwc4", _c_fp), ("V", _c_i), ("IH", _c_i), ("IW", _c_i),
        ("w2c", _c_fp), ("K", _c_fp), ("packed_mlp", _c_fp), ("packed_mlp_bf16", _c_fp),
        ("K_tgt", _c_fp), ("c2w_tgt", _c_fp), ("K_ref", _c_fp), ("w2c_ref", _c_fp), ("near_far_tgt", _c_fp), ("near_far_ref", _c_fp),
        ("W_img", _c_i), ("H_img", _c_i), ("pad", _c_i), ("lindisp", _c_i), ("W_ref", _c_i), ("H_ref", _c_i),
        ("first_pixel", _c_l), ("n_pixels", _c_l),
        ("S", _c_i), ("white_bkgd", _c_i), ("batch_rays", _c_i),
        ("workspace", _c_fp), ("workspace_floats", ctypes.c_size_t),
        ("rgb", _c_fp), ("depth", _c_fp), ("acc", _c_fp), ("disp", _c_fp), ("packed_mlp_split", _c_fp), ("n_split", _c_i), ("guard", _c_fp), ("vol_layout", _c_i),
    ]


class SweepConv0Args(ctypes.Structure):
    """mvsnerf_sweep_conv0_args (include/mvsnerf_hip.h): the guarded head of a no-grad scene encode."""
    _fields_ = [
        ("feats_cl", _c_fp), ("imgs_cl", _c_fp), ("proj", _c_fp), ("depth", _c_fp),
        ("V", _c_i), ("H", _c_i), ("W", _c_i), ("D", _c_i), ("pad", _c_i), ("CP", _c_i),
        ("masks", _c_fp), ("cost16x2", _c_fp), ("cost32", _c_fp), ("w_f16x3", _c_fp), ("w_c8", _c_fp), ("Cin", _c_i),
        ("out", _c_fp), ("stats_part", _c_fp), ("guard", _c_fp),
    ]


# name -> (restype, argtypes); must list every symbol of include/mvsnerf_hip.h (tests check this)
SIGNATURES = {
    "mvsnerf_abi_version": (_c_i, []),
    "mvsnerf_ncdhw_to_ndhwc": (_c_i, [_c_fp, _c_fp, _c_i, _c_i, _c_i, _c_i, _c_fp]),
    "mvsnerf_ndhwc_to_ncdhw": (_c_i, [_c_fp, _c_fp, _c_i, _c_i, _c_i, _c_i, _c_fp]),
    "mvsnerf_nchw_to_nhwc": (_c_i, [_c_fp, _c_fp, _c_i, _c_i, _c_i, _c_i, _c_i, _c_fp]),
    "mvsnerf_resize_bilinear": (_c_i, [_c_fp, _c_fp, _c_i, _c_i, _c_i, _c_i, _c_i, _c_fp]),
    "mvsnerf_planesweep_costvar_fwd": (_c_i, [_c_fp, _c_fp, _c_fp, _c_fp, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_fp, _c_i, _c_fp, _c_i, _c_fp]),
    "mvsnerf_planesweep_costvar_blocked_fwd": (_c_i, [_c_fp, _c_fp, _c_fp, _c_fp, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_fp, _c_i, _c_fp, _c_i, _c_fp]),
    "mvsnerf_planesweep_costvar_bf16_fwd": (_c_i, [_c_fp, _c_fp, _c_fp, _c_fp, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_fp, _c_i, _c_fp, _c_i, _c_fp]),
    "mvsnerf_conv0_bf16_packed_elems": (ctypes.c_size_t, [_c_i]),
    "mvsnerf_conv0_bf16_pack": (_c_i, [_c_fp, _c_i, _c_fp, _c_fp]),
    "mvsnerf_conv0_bf16_tiles": (_c_i, [_c_i] * 3),
    "mvsnerf_conv0_bf16_fwd": (_c_i, [_c_fp, _c_i, _c_i, _c_i, _c_i, _c_fp, _c_fp, _c_fp, _c_fp]),
    "mvsnerf_planesweep_costvar_f16x2_fwd": (_c_i, [_c_fp, _c_fp, _c_fp, _c_fp, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_fp, _c_i, _c_fp, _c_i, _c_fp]),
    "mvsnerf_conv0_f16x3_packed_elems": (ctypes.c_size_t, [_c_i]),
    "mvsnerf_conv0_f16x3_pack": (_c_i, [_c_fp, _c_i, _c_fp, _c_fp]),
    "mvsnerf_conv0_f16x3_fwd": (_c_i, [_c_fp, _c_i, _c_i, _c_i, _c_i, _c_fp, _c_fp, _c_fp, _c_fp]),
    "mvsnerf_conv0_bf16_dgrad_packed_elems": (ctypes.c_size_t, [_c_i]),
    "mvsnerf_conv0_bf16_dgrad_pack": (_c_i, [_c_fp, _c_i, _c_i, _c_i, _c_fp, _c_fp]),
    "mvsnerf_conv0_bf16_dgrad": (_c_i, [_c_fp, _c_i, _c_i, _c_i, _c_fp, _c_i, _c_fp, _c_fp]),
    "mvsnerf_conv0_bf16_wgrad_parts": (_c_i, [_c_i] * 3),
    "mvsnerf_conv0_bf16_wgrad": (_c_i, [_c_fp, _c_i, _c_i, _c_i, _c_i, _c_fp, _c_fp, _c_fp, _c_fp]),
    "mvsnerf_conv3d_bf16_packed_elems": (ctypes.c_size_t, [_c_i, _c_i, _c_i]),
    "mvsnerf_conv3d_bf16_pack": (_c_i, [_c_fp, _c_i, _c_i, _c_i, _c_fp, _c_fp]),
    "mvsnerf_conv3d_bf16_tiles": (_c_i, [_c_i] * 4),
    "mvsnerf_conv3d_bf16_fwd": (_c_i, [_c_fp] * 6 + [_c_i] * 5 + [_c_fp, _c_i, _c_i, _c_fp, _c_fp, _c_fp]),
    "mvsnerf_conv_transpose3d_bf16_tiles": (_c_i, [_c_i] * 3),
    "mvsnerf_conv_transpose3d_bf16_fwd": (_c_i, [_c_fp] * 6 + [_c_i] * 4 + [_c_fp, _c_i, _c_fp, _c_fp, _c_fp]),
    "mvsnerf_conv2d_bf16_packed_elems": (ctypes.c_size_t, [_c_i, _c_i, _c_i]),
    "mvsnerf_conv2d_bf16_tiles": (_c_i, [_c_i] * 5),
    "mvsnerf_conv2d_bf16_fwd": (_c_i, [_c_fp] * 3 + [_c_i] * 5 + [_c_fp, _c_fp] + [_c_i] * 3 + [_c_fp, _c_fp, _c_fp]),
    "mvsnerf_conv_wgrad_bf16_parts": (_c_i, [_c_i] * 8),
    "mvsnerf_conv_wgrad_bf16_workspace_floats": (ctypes.c_size_t, [_c_i] * 4),
    "mvsnerf_conv_wgrad_bf16": (_c_i, [_c_fp] * 6 + [_c_i] + [_c_fp] * 6 + [_c_i] * 11 + [_c_fp, _c_fp, _c_fp]),
    "mvsnerf_conv3d_mfma_supported": (_c_i, [_c_i, _c_i, _c_i]),
    "mvsnerf_conv3d_pack_weights_mfma": (_c_i, [_c_fp, _c_i, _c_i, _c_fp, _c_fp]),
    "mvsnerf_conv3d_mfma_fwd": (_c_i, [_c_fp, _c_fp, _c_fp, _c_i, _c_i, _c_i, _c_i, _c_i, _c_fp, _c_i, _c_i, _c_fp, _c_fp, _c_fp]),
    "mvsnerf_conv3d_mfma_tiles": (_c_i, [_c_i] * 4),
    "mvsnerf_conv_transpose3d_mfma_supported": (_c_i, [_c_i, _c_i]),
    "mvsnerf_conv_transpose3d_mfma_fwd": (_c_i, [_c_fp, _c_i, _c_i, _c_i, _c_i, _c_fp, _c_i, _c_fp, _c_fp]),
    "mvsnerf_conv3d_tiled_tiles": (_c_i, [_c_i, _c_i, _c_i]),
    "mvsnerf_conv3d_fwd_stats": (_c_i, [_c_fp, _c_fp, _c_fp, _c_i, _c_i, _c_i, _c_i, _c_i, _c_fp, _c_i, _c_i, _c_fp, _c_fp, _c_fp]),
    "mvsnerf_conv_transpose3d_mfma_fwd_stats": (_c_i, [_c_fp, _c_i, _c_i, _c_i, _c_i, _c_fp, _c_i, _c_fp, _c_fp, _c_fp]),
    "mvsnerf_conv_transpose3d_mfma_tiles": (_c_i, [_c_i, _c_i, _c_i, _c_i, _c_i]),
    "mvsnerf_conv_transpose3d_c8_supported": (_c_i, [_c_i, _c_i]),
    "mvsnerf_conv3d_c8_blocked_tiles": (_c_i, [_c_i] * 3),
    "mvsnerf_conv3d_c8_blocked_fwd_stats": (_c_i, [_c_fp, _c_i, _c_i, _c_i, _c_i, _c_i, _c_fp, _c_fp, _c_fp, _c_fp]),
    "mvsnerf_conv_transpose3d_c8_tiles": (_c_i, [_c_i] * 3),
    "mvsnerf_abn_finalize": (_c_i, [_c_fp, _c_i, _c_i, _c_l, _c_fp, _c_fp, _c_fp, _c_fp, ctypes.c_float, ctypes.c_float, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp]),
    "mvsnerf_conv_transpose3d_c8_fwd": (_c_i, [_c_fp] * 6 + [_c_i] * 4 + [_c_fp, _c_fp, _c_fp, _c_fp]),
    "mvsnerf_conv3d_pack_weights_c8": (_c_i, [_c_fp, _c_i, _c_fp, _c_fp]),
    "mvsnerf_conv3d_c8_blocked_fwd": (_c_i, [_c_fp, _c_i, _c_i, _c_i, _c_i, _c_i, _c_fp, _c_fp, _c_fp]),
    "mvsnerf_conv3d_c8_blocked_wgrad": (_c_i, [_c_fp, _c_i, _c_i, _c_i, _c_i, _c_i, _c_fp, _c_fp, _c_fp, _c_fp]),
    "mvsnerf_conv3d_c8_blocked_wgrad_parts": (_c_i, [_c_i] * 5),
    "mvsnerf_conv3d_wgrad_parts": (_c_i, [_c_i] * 7),
    "mvsnerf_conv2d_wgrad_parts": (_c_i, [_c_i] * 7),
    "mvsnerf_partial_sum_multi_scratch_floats": (ctypes.c_size_t, [_c_l]),
    "mvsnerf_partial_sum_multi": (_c_i, [_c_i, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp]),
    "mvsnerf_pack_weights_multi": (_c_i, [_c_i, _c_fp, _c_fp, _c_fp, _c_fp]),
    "mvsnerf_homo_warp_fwd": (_c_i, [_c_fp, _c_fp, _c_fp, _c_fp, _c_i, _c_i, _c_i, _c_i, _c_i, _c_fp, _c_fp, _c_fp]),
    "mvsnerf_conv3d_pack_weights": (_c_i, [_c_fp] + [_c_i] * 7 + [_c_fp, _c_fp]),
    "mvsnerf_conv3d_fwd": (_c_i, [_c_fp] * 6 + [_c_i] * 5 + [_c_fp, _c_i, _c_i, _c_fp, _c_fp]),
    "mvsnerf_conv_transpose3d_fwd": (_c_i, [_c_fp] * 6 + [_c_i] * 4 + [_c_fp, _c_i, _c_fp, _c_fp]),
    "mvsnerf_abn_workspace_floats": (ctypes.c_size_t, [_c_i]),
    "mvsnerf_abn_stats": (_c_i, [_c_fp, _c_l, _c_i, _c_fp, _c_fp, _c_fp, _c_fp, ctypes.c_float, ctypes.c_float, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp]),
    "mvsnerf_abn_bwd": (_c_i, [_c_fp, _c_l, _c_i] + [_c_fp] * 12),
    "mvsnerf_conv3d_wgrad_workspace_floats": (ctypes.c_size_t, [_c_i, _c_i]),
    "mvsnerf_conv3d_wgrad": (_c_i, [_c_fp] * 6 + [_c_i] + [_c_fp] * 6 + [_c_i] * 9 + [_c_fp, _c_fp, _c_fp]),
    "mvsnerf_planesweep_costvar_bwd": (_c_i, [_c_fp, _c_fp, _c_fp] + [_c_i] * 6 + [_c_fp, _c_i, _c_i, _c_fp, _c_fp]),
    "mvsnerf_adam_step_multi": (_c_i, [_c_i, ctypes.POINTER(_c_fp), ctypes.POINTER(_c_fp), ctypes.POINTER(_c_fp), ctypes.POINTER(_c_fp), ctypes.POINTER(ctypes.c_int64)]
                                + [ctypes.c_double] * 5 + [_c_fp]),
    "mvsnerf_planesweep_costvar_bwd_det_workspace_words": (ctypes.c_size_t, [_c_i] * 4),
    "mvsnerf_planesweep_costvar_bwd_det": (_c_i, [_c_fp, _c_fp, _c_fp] + [_c_i] * 6 + [_c_fp, _c_i, _c_i, _c_fp, _c_fp, _c_fp]),
    "mvsnerf_conv2d_pack_weights": (_c_i, [_c_fp] + [_c_i] * 8 + [_c_fp, _c_fp]),
    "mvsnerf_conv2d_fwd": (_c_i, [_c_fp] * 3 + [_c_i] * 5 + [_c_fp, _c_fp] + [_c_i] * 3 + [_c_fp, _c_fp]),
    "mvsnerf_conv2d_mfma_tiles": (_c_i, [_c_i] * 7),
    "mvsnerf_conv2d_fwd_stats": (_c_i, [_c_fp] * 3 + [_c_i] * 5 + [_c_fp] + [_c_i] * 3 + [_c_fp, _c_fp, _c_fp]),
    "mvsnerf_resize_bilinear_nhwc4": (_c_i, [_c_fp, _c_fp] + [_c_i] * 5 + [_c_fp]),
    "mvsnerf_depth_values": (_c_i, [_c_fp, _c_fp, _c_i, _c_fp, _c_fp]),
    "mvsnerf_conv2d_c3_nchw_fwd_stats": (_c_i, [_c_fp] + [_c_i] * 3 + [_c_fp, _c_fp, _c_fp, _c_fp]),
    "mvsnerf_conv2d_dgrad_k5s2": (_c_i, [_c_fp] + [_c_i] * 4 + [_c_fp] + [_c_i] * 3 + [_c_fp, _c_fp]),
    "mvsnerf_conv2d_wgrad_workspace_floats": (ctypes.c_size_t, [_c_i, _c_i, _c_i]),
    "mvsnerf_conv2d_wgrad": (_c_i, [_c_fp, _c_i] + [_c_fp] * 3 + [_c_i] * 9 + [_c_fp, _c_fp, _c_fp]),
    "mvsnerf_channel_sum_workspace_floats": (ctypes.c_size_t, [_c_i]),
    "mvsnerf_channel_sum": (_c_i, [_c_fp, _c_l, _c_i, _c_fp, _c_fp, _c_fp]),
    "mvsnerf_sample_pdf_fwd": (_c_i, [_c_fp, _c_fp, _c_fp, _c_l, _c_i, _c_i, _c_fp, _c_fp]),
    "mvsnerf_ray_marcher_fine_fwd": (_c_i, [_c_fp, _c_i, _c_i, _c_i, _c_fp, _c_fp, _c_fp, _c_l, _c_i, _c_i, _c_fp, _c_fp]),
    "mvsnerf_ray_points_fwd": (_c_i, [_c_fp, _c_i, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_i, _c_i, _c_i, _c_i, _c_l, _c_i, _c_fp, _c_fp, _c_fp]),
    "mvsnerf_render_workspace_floats": (ctypes.c_size_t, [_c_i, _c_i, _c_i]),
    "mvsnerf_render_pixels_fwd": (_c_i, [ctypes.POINTER(RenderArgs), _c_fp]),
    "mvsnerf_gather_fwd": (_c_i, [_c_fp] + [_c_i] * 3 + [_c_fp] + [_c_i] * 3 + [_c_fp] * 4 + [_c_l, _c_i, _c_fp, _c_fp, _c_i, _c_fp, _c_i, _c_fp]),
    "mvsnerf_abn_apply_add": (_c_i, [_c_fp] * 6 + [_c_l, _c_i, _c_fp, _c_fp]),
    "mvsnerf_abn_apply_add_hwdc": (_c_i, [_c_fp] * 6 + [_c_i, _c_i, _c_i, _c_fp, _c_fp]),
    "mvsnerf_raygen_fwd": (_c_i, [_c_fp, _c_fp, _c_l, _c_i, _c_i, _c_i, _c_i] + [_c_fp] * 6 + [_c_i, _c_i, _c_fp, _c_l, _c_i] + [_c_fp] * 6),
    "mvsnerf_raygen_train_fwd": (_c_i, [_c_fp, _c_fp, _c_i, _c_i, _c_i, _c_i] + [_c_fp] * 6 + [_c_i, _c_i, _c_fp, _c_l, _c_i] + [_c_fp] * 3 + [_c_i] + [_c_fp] * 8),
    "mvsnerf_volume_sample_fwd": (_c_i, [_c_fp, _c_i, _c_i, _c_i, _c_i, _c_fp, _c_l, _c_fp, _c_i, _c_i, _c_fp]),
    "mvsnerf_color_sample_fwd": (_c_i, [_c_fp, _c_i, _c_i, _c_i, _c_fp, _c_fp, _c_fp, _c_l, _c_i, _c_fp, _c_i, _c_fp]),
    "mvsnerf_color_feat_sample_fwd": (_c_i, [_c_fp, _c_i, _c_i, _c_i, _c_fp, _c_i, _c_i, _c_i, _c_fp, _c_fp, _c_fp, _c_l, _c_i, _c_fp, _c_i, _c_fp]),
    "mvsnerf_dir_feature_fwd": (_c_i, [_c_fp, _c_fp, _c_l, _c_i, _c_fp, _c_fp]),
    "mvsnerf_posenc_fwd": (_c_i, [_c_fp, _c_l, _c_i, _c_i, _c_fp, _c_fp]),
    "mvsnerf_mlp_packed_floats": (ctypes.c_size_t, [_c_i]),
    "mvsnerf_mlp_pack": (_c_i, [ctypes.POINTER(_c_fp), ctypes.POINTER(_c_fp), _c_i, _c_fp, _c_fp]),
    "mvsnerf_mlp_fwd": (_c_i, [_c_fp, _c_i, _c_fp, _c_i, _c_fp, _c_i, _c_fp, _c_i, _c_l, _c_i, _c_i, _c_fp, _c_fp]),
    "mvsnerf_mlp_fwd_census": (_c_i, [_c_fp, _c_i, _c_fp, _c_i, _c_fp, _c_i, _c_fp, _c_i, _c_l, _c_i, _c_i, _c_fp, _c_fp, _c_fp]),
    "mvsnerf_mlp_packed_split_elems": (ctypes.c_size_t, [_c_i, _c_i]),
    "mvsnerf_mlp_pack_split": (_c_i, [ctypes.POINTER(_c_fp), _c_i, _c_i, _c_fp, _c_fp]),
    "mvsnerf_mlp_fwd_split": (_c_i, [_c_fp, _c_fp, _c_i, _c_i, _c_fp, _c_i, _c_fp, _c_i, _c_fp, _c_i, _c_l, _c_i, _c_i, _c_fp, _c_fp]),
    "mvsnerf_mlp_fwd_guarded": (_c_i, [_c_fp, _c_fp, _c_i, _c_fp, _c_i, _c_fp, _c_i, _c_fp, _c_i, _c_l, _c_i, _c_i, _c_fp, _c_fp, _c_fp]),
    "mvsnerf_sweep_conv0_guarded_fwd": (_c_i, [ctypes.POINTER(SweepConv0Args), _c_fp]),
    "mvsnerf_conv3d_f16x3_supported": (_c_i, [_c_i, _c_i, _c_i]),
    "mvsnerf_conv3d_f16x3_slots": (_c_i, []),
    "mvsnerf_conv3d_f16x3_packed_elems": (ctypes.c_size_t, [_c_i]),
    "mvsnerf_conv3d_f16x3_pack": (_c_i, [_c_fp, _c_i, _c_i, _c_fp, _c_fp]),
    "mvsnerf_conv3d_f16x3_fwd": (_c_i, [_c_fp, _c_fp, _c_fp, _c_i, _c_i, _c_i, _c_i, _c_i, _c_fp, _c_i, _c_i, _c_fp, _c_fp, _c_fp]),
    "mvsnerf_conv3d_f16x3_guarded_fwd": (_c_i, [_c_fp, _c_fp, _c_fp, _c_i, _c_i, _c_i, _c_i, _c_i, _c_fp, _c_fp, _c_i, _c_i, _c_fp, _c_fp, _c_fp, _c_i, _c_fp]),
    "mvsnerf_mlp_packed_bf16_elems": (ctypes.c_size_t, [_c_i]),
    "mvsnerf_mlp_pack_bf16": (_c_i, [ctypes.POINTER(_c_fp), _c_i, _c_fp, _c_fp]),
    "mvsnerf_mlp_fwd_bf16": (_c_i, [_c_fp, _c_fp, _c_i, _c_fp, _c_i, _c_fp, _c_i, _c_fp, _c_i, _c_l, _c_i, _c_i, _c_fp, _c_fp]),
    "mvsnerf_mlp_saved_floats": (ctypes.c_size_t, [_c_l]),
    "mvsnerf_mlp_gradslot_floats": (ctypes.c_size_t, [_c_l]),
    "mvsnerf_mlp_packed_bwd_floats": (ctypes.c_size_t, []),
    "mvsnerf_mlp_bwd_workspace_floats": (ctypes.c_size_t, []),
    "mvsnerf_mlp_pack_bwd": (_c_i, [ctypes.POINTER(_c_fp), _c_i, _c_fp, _c_fp]),
    "mvsnerf_mlp_fwd_train": (_c_i, [_c_fp, _c_i, _c_fp, _c_i, _c_fp, _c_i, _c_fp, _c_i, _c_l, _c_i, _c_fp, _c_fp, _c_fp]),
    "mvsnerf_mlp_fwd_bf16_train": (_c_i, [_c_fp, _c_fp, _c_i, _c_fp, _c_i, _c_fp, _c_i, _c_fp, _c_i, _c_l, _c_i, _c_fp, _c_fp, _c_fp]),
    "mvsnerf_mlp_packed_bwd_bf16_elems": (ctypes.c_size_t, []),
    "mvsnerf_mlp_pack_bwd_bf16": (_c_i, [ctypes.POINTER(_c_fp), _c_i, _c_fp, _c_fp]),
    "mvsnerf_mlp_bwd_bf16": (_c_i, [_c_fp, _c_fp, _c_i, _c_fp, _c_fp, _c_fp, _c_l, _c_i, _c_fp, _c_fp, _c_i, ctypes.POINTER(_c_fp), ctypes.POINTER(_c_fp), _c_fp, _c_fp, _c_fp]),
    "mvsnerf_mlp_bwd": (_c_i, [_c_fp, _c_fp, _c_i, _c_fp, _c_fp, _c_fp, _c_l, _c_i, _c_fp, _c_fp, _c_i, ctypes.POINTER(_c_fp), ctypes.POINTER(_c_fp), _c_fp, _c_fp, _c_fp]),
    "mvsnerf_composite_bwd": (_c_i, [_c_fp, _c_fp, _c_l, _c_i, _c_i, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp]),
    "mvsnerf_volume_sample_bwd": (_c_i, [_c_i, _c_i, _c_i, _c_i, _c_fp, _c_l, _c_fp, _c_i, _c_fp, _c_fp]),
    "mvsnerf_volume_sample_bwd_det_workspace_words": (ctypes.c_size_t, [_c_i] * 4),
    "mvsnerf_volume_sample_bwd_det": (_c_i, [_c_i, _c_i, _c_i, _c_i, _c_fp, _c_l, _c_fp, _c_i, _c_fp, _c_fp, _c_fp]),
    "mvsnerf_composite_fwd": (_c_i, [_c_fp, _c_fp, _c_l, _c_i, _c_i, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp]),
    "mvsnerf_raymarch_fwd": (_c_i, [ctypes.POINTER(RaymarchArgs), _c_fp]),
    "mvsnerf_raymarch_fwd_batched": (_c_i, [ctypes.POINTER(RaymarchArgs), _c_i, _c_fp]),
    "mvsnerf_raymarch_train_fwd": (_c_i, [ctypes.POINTER(RaymarchTrainArgs), _c_fp]),
    "mvsnerf_raymarch_bwd": (_c_i, [ctypes.POINTER(RaymarchBwdArgs), _c_fp]),
}

_lib = None


def build(verbose=False):
    """Compile libmvsnerf_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    r = subprocess.run(["make", "-C", CSRC, "-j8"], capture_output=True, text=True)
    if verbose or r.returncode:
        print(r.stdout[-4000:], r.stderr[-4000:])
    if r.returncode:
        raise RuntimeError("building libmvsnerf_hip.so failed")
    return LIB_PATH


def lib():
    """The loaded library; raises loudly when it is absent (no CPU / eager fallback exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the MVSNeRF hot path has no fallback. Build it with "
                "`python -c 'import __graft_entry__ as g; g.build()'` or `make -C mvsnerf_amd/csrc`.")
        l = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)          # AttributeError if the .so lacks a declared symbol
            fn.restype, fn.argtypes = res, args
        _lib = l
    return _lib


def stream_ptr():
    return torch.cuda.current_stream().cuda_stream


def check(rc, op):
    if rc != 0:
        kind = {-1: "invalid argument", -2: "unsupported shape", -3: "misaligned pointer"}.get(rc, f"hipError {rc}")
        raise RuntimeError(f"libmvsnerf_hip: {op} failed: {kind}")


def dev_f32(t, name, cur=None):
    """Validate a tensor handed to the C ABI: CUDA(HIP) device, fp32, contiguous.  cur: torch.cuda.current_device() when the caller has
    already asked for it (one query per FFI call instead of one per tensor: the ray-march step is paced by this host path)."""
    if not (torch.is_tensor(t) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise RuntimeError(f"{name}: expected a contiguous float32 tensor on the GPU, got "
                           f"{type(t).__name__} {getattr(t, 'dtype', None)} {getattr(t, 'device', None)}")
    if t.device.index != (torch.cuda.current_device() if cur is None else cur):
        # kernels are launched on the CURRENT device's stream (stream_ptr): a tensor living on another GPU would be dereferenced
        # by the wrong device.  One process per GPU with torch.cuda.set_device(LOCAL_RANK) is the supported layout.
        raise RuntimeError(f"{name}: tensor is on {t.device} but the current device is cuda:{torch.cuda.current_device()} "
                           "(call torch.cuda.set_device / use `with torch.cuda.device(...)` around the call)")
    return t.data_ptr()


# ---- weight-cache epoch.  The packed-weight caches (models.MVSNeRF.packed*, encoder._PackedConv*) key on (data_ptr, tensor._version),
# but a FUSED optimizer (`torch.optim.Adam(..., fused=True)`) updates the parameters without bumping `_version` (checked on torch 2.10:
# version 0 -> 0), and a training loop would silently keep running on the first step's packed weights.  Every optimizer step anywhere
# in the process therefore advances this counter, and it is part of every cache key.  (Writes through `.data` are still invisible:
# invalidate_packed().)
_WEIGHTS_EPOCH = [0]


def _bump_weights_epoch(*_a, **_k):
    _WEIGHTS_EPOCH[0] += 1


try:
    from torch.optim.optimizer import register_optimizer_step_post_hook as _reg_hook
    _reg_hook(_bump_weights_epoch)
except Exception:                                   # very old torch: no global hook; fused optimizers then need invalidate_packed()
    pass


def weights_epoch():
    return _WEIGHTS_EPOCH[0]

