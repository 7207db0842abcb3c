"""ctypes binding of ``libstroke_amd.so`` (the C ABI in ``include/stroke_amd.h``).

The library is built in-tree by ``__graft_entry__.build()`` (``hipcc
--offload-arch=gfx950``).  There is no fallback: if it is missing or a call
fails, a ``RuntimeError`` carrying ``sp_last_error`` is raised.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
PKG_DIR = os.path.dirname(_HERE)
LIB_PATH = os.environ.get("SP_LIB_PATH") or os.path.join(PKG_DIR, "lib", "libstroke_amd.so")   # SP_LIB_PATH: diagnostic builds (tools/)
CSRC_DIR = os.path.join(PKG_DIR, "csrc")
SOURCES = ["sp_conv.hip", "sp_conv_dma.hip", "sp_conv_par.hip", "sp_conv_zm.hip", "sp_conv_zm8.hip", "sp_wgrad.hip", "sp_wgrad_dma.hip", "sp_conv_fc.hip", "sp_wgrad_zr.hip", "sp_wgrad_pw.hip", "sp_wgrad_f8.hip", "sp_plan.hip", "sp_comm.hip", "sp_head.hip", "sp_first.hip", "sp_elem.hip", "sp_pwout.hip",
           "sp_transform.hip"]

SP_BF16, SP_F32, SP_HL = 0, 1, 2      # SP_HL: bf16 pair (hi + lo tensors), the forward storage of the "bf16x3" mode
# precision modes of the models (``Unet3D(dtype=...)``, ``Enc3D(dtype=...)``) -> storage type of the engine's tensors
DTYPE_CODES = {"bf16": SP_BF16, "f32": SP_F32, "fp8": SP_BF16, "fp8b": SP_BF16, "f16": SP_BF16, "bf16x3": SP_BF16, "f16x3": SP_BF16}
#   fp8: bf16 storage + fp8 MFMA operands (runtime/f8.py); fp8b: the bf16 forward with the fp8 BACKWARD (data and weight
#   gradients on e5m2 / e4m3 operands) -- the forward, and with it the direction of the gradients, is the bf16 mode's; f16: IEEE-half storage -- the SAME sources built with
#   -DSP_HALF_F16 into libstroke_amd_f16.so (csrc/sp_common.h), selected per engine with ``use("f16")``; the kernels'
#   dtype code stays SP_BF16 = "the 16-bit storage type of this library";
#   bf16x3: the FORWARD activations are bf16 pairs (hi + lo tensors, SP_HL: ~17 bits; three MFMAs per product), the backward
#   pass is the bf16 one on the hi tensors -- logits within 1e-3 of the fp32 reference at ~1.5x the bf16 step;
#   f16x3: the same in the IEEE-half build (pairs of halves: ~22 bits forward; the backward is the f16 mode's, 8x closer than bf16)
VARIANTS = {"": ("libstroke_amd.so", []), "f16": ("libstroke_amd_f16.so", ["-DSP_HALF_F16"])}
VARIANT_OF = {"bf16": "", "f32": "", "fp8": "", "fp8b": "", "f16": "f16", "bf16x3": "", "f16x3": "f16"}
SP_REDUCE_ROWS = 8    # replica rows of the accumulators the elementwise kernels reduce into (include/stroke_amd.h)
ACT_NONE, ACT_LEAKY, ACT_ELU, ACT_SIGMOID = 0, 1, 2, 3

i32, i64, f32, f64, vp = C.c_int32, C.c_int64, C.c_float, C.c_double, C.c_void_p


class BnFinArgs(C.Structure):      # sp_bn_fin_args
    _fields_ = [(n, vp) for n in ("sums", "gamma", "beta", "running_mean", "running_var", "scale", "shift", "mean", "invstd")] + \
               [("count", f64), ("momentum", f32), ("eps", f32), ("nrep", i32), ("training", i32), ("C", i32), ("CP", i32)]


class BnBwdArgs(C.Structure):      # sp_bn_bwd_args
    _fields_ = [(n, vp) for n in ("sums", "gamma", "mean", "invstd", "dgamma", "dbeta", "coef")] + \
               [("count", f64), ("pscale", f32), ("nrep", i32), ("C", i32), ("CP", i32)]


class ConvArgs(C.Structure):
    _fields_ = [(n, vp) for n in ("x", "y", "wfrag_hi", "wfrag_lo", "in_scale", "in_shift", "bias", "stats", "ktab")] + \
               [(n, i32) for n in (
                   "dtype_in", "dtype_out", "B", "Di", "Hi", "Wi", "CPi", "Do", "Ho", "Wo", "YD", "YH", "YW", "CPo",
                   "osD", "osH", "osW", "ooD", "ooH", "ooW", "Cout", "sD", "sH", "sW", "o0D", "o0H", "o0W",
                   "TD", "TH", "ITD", "ITH", "ITW", "MT", "NT", "NTtot", "ngroups", "octs_per_group", "opp", "vsb",
                   "plane_bytes", "lo_offset", "steps_per_group", "lds_bytes", "act")] + [("act_param", f32), ("dma", i32), ("zfill", i32), ("persist", i32), ("aux", vp), ("stats_mode", i32), ("stats_nrep", i32), ("ITH_zs", i32), ("x_plane", i64),
                                                                           ("y8", vp), ("y8_plane", i64), ("f8_wscale", vp), ("y8_scale", f32), ("f8_bin", i32),
                                                                           ("group_batch", i32), ("nslices", i32),
                                                                          ("slice_wfrag_stride", i64), ("x_lo_delta", i64), ("y_lo_delta", i64),
                                                                          ("bias_tab", vp), ("bias_tab_gstride", i32), ("pad_", i32), ("wfrag_gstride", i64),
                                                                          ("bnb", BnBwdArgs), ("dz_sums", vp), ("pool_y", vp), ("pool_lo_delta", i64), ("y2", vp), ("split_nt", i32), ("CPo2", i32), ("pser_planes", i32), ("pad2_", i32)]


class WgradArgs(C.Structure):
    _fields_ = [(n, vp) for n in ("x", "dz", "in_scale", "in_shift", "dz_scale", "dz_shift", "dw_acc", "taps")] + \
               [(n, i32) for n in ("dtype", "B", "Di", "Hi", "Wi", "CPi", "Do", "Ho", "Wo", "CPo", "sD", "sH", "sW",
                                   "o0D", "o0H", "o0W", "ntap", "kD", "kH", "kW", "CoT", "CiT", "nblocks", "dma", "tile_rows", "parts", "cib")] + [("x_plane", i64), ("zs", i32), ("groups", i32)]


class WgradF8Args(C.Structure):      # sp_wgrad_f8_args
    _fields_ = [(n, vp) for n in ("x", "dz", "dw_acc")] + \
               [(n, i32) for n in ("B", "Di", "Hi", "Wi", "Do", "Ho", "Wo", "CoT", "CiT", "nblocks")] + [("x_plane", i64), ("dz_plane", i64)]


class ConvFcArgs(C.Structure):       # sp_conv_fc_args
    _fields_ = [(n, vp) for n in ("x", "y", "wfrag", "in_scale", "in_shift", "bias", "stats", "aux", "partial", "taps")] + \
               [(n, i32) for n in ("B", "Di", "Hi", "Wi", "CPi", "Do", "Ho", "Wo", "CPo", "Cout", "sD", "sH", "sW", "o0D", "o0H", "o0W",
                                   "ntap", "act")] + [("act_param", f32), ("stats_mode", i32), ("stats_nrep", i32), ("dtype_out", i32), ("x_plane", i64),
                                                                         ("group_batch", i32), ("coef_gstride", i32)]


class Conv3dDesc(C.Structure):       # sp_conv3d_desc
    _fields_ = [(n, i32) for n in ("B", "Cin", "Cout", "D", "H", "W", "grad", "padD", "padH", "padW", "transposed")]


class Conv3dPlan(C.Structure):       # sp_conv3d_plan_t
    _fields_ = [(n, i32) for n in ("cin_op", "cout_op", "P", "NT", "MT", "NW", "NSLOT", "KS", "nsteps", "ITH",
                                   "Di", "Hi", "Wi", "Do", "Ho", "Wo", "o0", "o0H", "o0W", "mirror")] + \
               [(n, i64) for n in ("x_elems", "y_elems", "workspace_bytes", "off_zero", "off_ktab", "off_kmap", "off_bias", "off_wfrag")]


class Conv3dWgradPlan(C.Structure):  # sp_conv3d_wgrad_plan_t
    _fields_ = [(n, i32) for n in ("CoT", "CiT", "nblocks", "Do", "Ho", "Wo")] + \
               [(n, i64) for n in ("workspace_bytes", "off_taps", "off_tapsrc", "off_acc")]


_SIGS = {
    "sp_conv_fc_workspace": ([i32, i32, i32, i32, i32, i32, C.POINTER(i64)], i32),
    "sp_conv_fc": ([C.POINTER(ConvFcArgs), vp], i32),
    "sp_conv3d_plan": ([C.POINTER(Conv3dDesc), C.POINTER(Conv3dPlan)], i32),
    "sp_conv3d_tables": ([C.POINTER(Conv3dDesc), C.POINTER(Conv3dPlan), vp, vp], i32),
    "sp_conv3d_init": ([C.POINTER(Conv3dDesc), C.POINTER(Conv3dPlan), vp, vp], i32),
    "sp_conv3d_set_weights": ([C.POINTER(Conv3dDesc), C.POINTER(Conv3dPlan), vp, vp, vp, vp, vp, vp], i32),
    "sp_conv3d_run": ([C.POINTER(Conv3dDesc), C.POINTER(Conv3dPlan), vp, vp, vp, i32, i32, f32, vp, i32, i64, vp], i32),
    "sp_conv3d_wgrad_plan": ([C.POINTER(Conv3dDesc), C.POINTER(Conv3dWgradPlan)], i32),
    "sp_conv3d_wgrad_init": ([C.POINTER(Conv3dDesc), C.POINTER(Conv3dWgradPlan), vp, vp], i32),
    "sp_conv3d_wgrad_run": ([C.POINTER(Conv3dDesc), C.POINTER(Conv3dWgradPlan), vp, vp, vp, vp, vp, vp, vp, i32, vp, vp, vp, i32, i64, vp], i32),
    "sp_version": ([], i32),
    "sp_comm_available": ([], i32),
    "sp_comm_unique_id": ([vp], i32),
    "sp_comm_init_rank": ([C.POINTER(vp), i32, vp, i32], i32),
    "sp_comm_destroy": ([vp], i32),
    "sp_allreduce_flat": ([vp, vp, i64, vp], i32),
    "sp_allreduce_flat_f64": ([vp, vp, i64, vp], i32),
    "sp_reduce_scatter_flat": ([vp, vp, i64, i32, vp], i32),
    "sp_allgather_flat": ([vp, vp, i64, i32, vp], i32),
    "sp_surface_distances": ([vp, vp, f32, i32, vp, vp, vp, vp], i32),
    "sp_gaussian_filter3d": ([vp, vp, vp, i32, i32, i32, f32, f32, vp], i32),
    "sp_map_coordinates_linear": ([vp, vp, vp, vp, f32, f32, f32, f32, vp, i32, i32, i32, vp], i32),
    "sp_conv_prep_weights_batch": ([vp, i32, i32, vp], i32),
    "sp_conv3d_igemm": ([C.POINTER(ConvArgs), vp], i32),
    "sp_conv3d_igemm_multi": ([C.POINTER(ConvArgs), i32, vp], i32),
    "sp_bn_act_bwd_groups_cls": ([vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, f32, vp, vp, i32, i32, i32, i32, vp, vp], i32),
    "sp_wgrad_finish_folded_groups": ([vp, i32, i32, i32, i32, i32, i32, i32, i64, i64, vp, i32, i32, vp, i32, i32, i32, vp, vp, vp, vp, i32, i32, vp], i32),
    "sp_cae_loss_fwd": ([vp, i64, vp, i64, vp, i64, vp, i64, vp, i64, vp, i64, vp, i64, i32, i64, vp, vp, i64, f32, f64, f32, vp, vp, vp, vp], i32),
    "sp_cae_loss_bwd": ([vp, i64, vp, i64, vp, i64, vp, i64, vp, i64, vp, i64, vp, i64, i32, i64, vp, vp, vp, vp, vp, vp, vp, vp, i64, vp, vp, vp], i32),
    "sp_pwout_fwd": ([vp, i32, i64, i32, i32, vp, i32, i32, vp, vp, vp, vp], i32),
    "sp_pwout_bwd": ([vp, vp, vp, i32, i64, i32, i32, vp, i32, i32, vp, vp, vp], i32),
    "sp_pwout_finish": ([vp, i32, i32, i32, vp, vp, i32, vp, i32, vp, vp, vp], i32),
    "sp_conv_prep_folded_groups": ([vp, i64, i64, i32, i32, vp, i32, i32, vp, i64, vp, i32, i32, i32, vp, i32, i32, i32, vp, i32, vp], i32),
    "sp_conv3d_par": ([C.POINTER(ConvArgs), i32, vp, C.POINTER(i32), vp, vp], i32),
    "sp_conv3d_zm": ([C.POINTER(ConvArgs), vp, vp], i32),
    "sp_conv3d_zm_config": ([i32, i32, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)], i32),
    "sp_conv3d_zm_config_hl": ([i32, i32, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)], i32),
    "sp_conv3d_zm_config_ps": ([i32, i32, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)], i32),
    "sp_bn_stats_ncdhw_f32": ([vp, i32, i32, i64, i32, vp, i32, vp], i32),
    "sp_first_prep_hl": ([vp, vp, vp, vp, vp, vp, vp, i32, vp], i32),
    "sp_first_conv_fwd_hl": ([vp, i32, i32, i32, i32, vp, vp, vp, i32, f32, vp, vp, vp, i32, i32, vp], i32),
    "sp_maxpool2_fwd_hl": ([vp, i64, vp, i64, i32, i32, i32, i32, i32, vp, vp], i32),
    "sp_upsample2_crop_cat_fwd_hl": ([vp, i64, i32, vp, i64, i32, vp, i64, i32, i32, i32, i32, i32, i32, i32, i32, i64, vp, vp], i32),
    "sp_head_fwd_hl": ([vp, i64, i64, i32, i32, i32, vp, vp, i32, vp, vp, i32, f32, vp, vp], i32),
    "sp_conv3d_zm8": ([C.POINTER(ConvArgs), vp, vp], i32),
    "sp_conv3d_zm8_config": ([i32, i32, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)], i32),
    "sp_conv_prep_f8": ([vp, i64, i64, i32, i32, vp, i32, i32, vp, vp, vp, i32, vp, vp, vp, f32, vp], i32),
    "sp_quantize_f8": ([vp, i32, i64, vp, i64, i64, i32, f32, vp], i32),
    "sp_conv_prep_f8_batch": ([vp, i32, i32, vp], i32),
    "sp_conv_prep_weights": ([vp, i64, i64, i32, i32, vp, i32, i32, vp, vp, vp, vp], i32),
    "sp_conv_fold_bias": ([vp, i64, i64, i32, i32, i32, vp, vp, vp, i32, vp], i32),
    "sp_conv_prep_folded": ([vp, i64, i64, i32, i32, vp, i32, i32, vp, vp, vp, i32, vp, vp, vp, i32, vp], i32),
    "sp_conv_prep_folded_bn": ([vp, i64, i64, i32, i32, vp, i32, i32, vp, vp, i32, vp, vp, i32, C.POINTER(BnFinArgs), vp], i32),
    "sp_first_prep_bn": ([vp, vp, vp, vp, vp, i32, C.POINTER(BnFinArgs), vp], i32),
    "sp_conv3d_wgrad": ([C.POINTER(WgradArgs), vp], i32),
    "sp_wgrad_finish": ([vp, i32, vp, i32, i32, i32, i32, i32, i64, i64, vp, vp, vp, i32, i32, vp], i32),
    "sp_wgrad_finish_folded": ([vp, i32, vp, i32, i32, i32, i32, i32, i64, i64, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp], i32),
    "sp_wgrad_finish_folded_scaled": ([vp, i32, vp, i32, i32, i32, i32, i32, i64, i64, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, f32, vp], i32),
    "sp_conv3d_wgrad_f8": ([C.POINTER(WgradF8Args), vp], i32),
    "sp_conv_partial_finish": ([vp, i32, i64, i32, vp, i32, i32, f32, vp, vp, i32, vp, i64, vp], i32),
    "sp_upsample2_crop_cat_fwd": ([vp, i32, vp, i32, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i64, vp, vp], i32),
    "sp_confusion_counts": ([vp, vp, f32, i64, vp, vp], i32),
    "sp_first_supported": ([i32, i32, i32], i32),
    "sp_bn_stats_ncdhw": ([vp, i32, i32, i64, i32, vp, i32, vp], i32),
    "sp_first_prep": ([vp, vp, vp, vp, vp, vp, vp], i32),
    "sp_first_conv_fwd": ([vp, i32, i32, i32, i32, vp, vp, i32, f32, vp, vp, i32, vp], i32),
    "sp_first_wgrad": ([vp, vp, i32, i32, i32, i32, vp, i32, vp], i32),
    "sp_first_wgrad_fused": ([vp, vp, vp, vp, i32, f32, i32, i32, i32, i32, vp, i32, vp, vp], i32),
    "sp_first_prep_n": ([vp, vp, vp, vp, vp, vp, i32, vp], i32),
    "sp_first_conv_fwd_n": ([vp, i32, i32, i32, i32, vp, vp, i32, f32, vp, vp, i32, i32, vp, i64, vp], i32),
    "sp_first_wgrad_n": ([vp, vp, i32, i32, i32, i32, vp, i32, i32, vp], i32),
    "sp_first_wgrad_fused_n": ([vp, vp, vp, vp, i32, f32, i32, i32, i32, i32, vp, i32, vp, i32, vp], i32),
    "sp_first_wgrad_fused_y8": ([vp, vp, vp, i64, vp, i32, f32, i32, i32, i32, i32, vp, i32, vp, i32, vp], i32),
    "sp_ncdhw_to_cl": ([vp, vp, i32, i32, i32, i64, i32, vp], i32),
    "sp_cl_to_ncdhw": ([vp, vp, i32, i32, i32, i64, i32, vp], i32),
    "sp_bn_stats": ([vp, i32, i64, i32, vp, vp], i32),
    "sp_bn_finalize": ([vp, i32, f64, vp, vp, vp, vp, f32, f32, i32, i32, i32, vp, vp, vp, vp, vp], i32),
    "sp_bn_bwd_reduce": ([vp, vp, i32, i64, i32, vp, vp], i32),
    "sp_bn_finalize_groups": ([vp, i32, f64, vp, vp, vp, vp, f32, f32, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp], i32),
    "sp_bn_bwd_finalize_groups": ([vp, i32, f64, vp, vp, vp, i32, i32, i32, vp, vp, vp, f32, vp], i32),
    "sp_bn_act_bwd_groups": ([vp, vp, vp, i32, i64, i32, i32, f32, vp, vp, i64, vp], i32),
    "sp_bn_bwd_finalize": ([vp, i32, f64, vp, vp, vp, i32, i32, vp, vp, vp, f32, vp], i32),
    "sp_bn_act_bwd": ([vp, vp, vp, i32, i64, i32, i32, f32, vp, vp, vp], i32),
    "sp_maxpool2_fwd": ([vp, vp, i32, i32, i32, i32, i32, i32, vp, vp], i32),
    "sp_maxpool2_fwd_q8": ([vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, i64, i32, f32, vp], i32),
    "sp_bn_act_bwd_q8": ([vp, vp, vp, i32, i64, i32, i32, f32, vp, vp, vp, i64, i32, f32, vp], i32),
    "sp_bn_act_bwd_y8": ([vp, vp, i64, vp, i64, i32, i32, f32, vp, vp, vp, i64, i32, f32, vp], i32),
    "sp_maxpool2_fwd_x8": ([vp, i64, vp, i32, i32, i32, i32, i32, vp, vp, i64, i32, f32, vp], i32),
    "sp_upsample2_crop_cat_fwd_q8s8": ([vp, i32, vp, i64, i32, vp, i32, i32, i32, i32, i32, i32, i32, i32, i64, vp, vp, i64, i32, f32, vp], i32),
    "sp_pool_skip_act_bwd_y8": ([vp, i64, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, f32, vp, vp, vp, i64, i32, f32, vp], i32),
    "sp_upsample2_crop_cat_fwd_q8": ([vp, i32, vp, i32, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i64, vp, vp, i64, i32, f32, vp], i32),
    "sp_pool_skip_act_bwd_q8": ([vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32,
                                 f32, vp, vp, vp, i64, i32, f32, vp], i32),
    "sp_upsample2_fwd": ([vp, vp, i32, i32, i32, i32, i32, i32, i32, vp, vp], i32),
    "sp_crop_copy": ([vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp], i32),
    "sp_pool_skip_act_bwd": ([vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32,
                              f32, vp, vp, vp], i32),
    "sp_upsample2_act_bwd": ([vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, f32, vp, vp, vp], i32),
    "sp_upsample2_act_bwd_q8": ([vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, f32, vp, vp, vp, i64, i32, f32, vp], i32),
    "sp_out_grad_to_cl": ([vp, vp, i32, i32, i64, i32, i32, i32, f32, vp, vp, vp], i32),
    "sp_dice_sums": ([vp, i64, vp, i64, i32, i32, i64, vp, vp], i32),
    "sp_dice_finalize": ([vp, vp, f64, i32, vp, vp, vp], i32),
    "sp_dice_finalize_clear": ([vp, vp, f64, i32, vp, vp, vp], i32),
    "sp_dice_bwd": ([vp, i64, vp, i64, vp, vp, i32, i32, i64, vp, vp], i32),
    "sp_head_supported": ([i32, i32, i32], i32),
    "sp_head_supported_dtype": ([i32, i32, i32, i32], i32),
    "sp_head_fwd": ([vp, i32, i64, i32, i32, i32, vp, vp, i32, vp, vp, i32, f32, vp, vp], i32),
    "sp_head_bwd_rows": ([i64], i64),
    "sp_head_row_floats": ([i32, i32, i32], i32),
    "sp_head_bwd": ([vp, i32, i64, i32, i32, i32, vp, vp, i32, vp, i32, f32, vp, vp, i32, f32, vp, vp, vp], i32),
    "sp_head_bwd_q8": ([vp, i32, i64, i32, i32, i32, vp, vp, i32, vp, i32, f32, vp, vp, i32, f32, vp, vp, vp, i64, i32, f32, vp], i32),
    "sp_head_grad_finish": ([vp, i64, i32, i32, i32, vp, vp, vp, vp, vp, vp], i32),
    "sp_add_f64_to_f32": ([vp, vp, i64, f32, vp], i32),
    "sp_axpby": ([vp, vp, vp, i32, i64, f32, f32, vp], i32),
    "sp_lerp_batch": ([vp, vp, vp, vp, i32, i32, i64, vp], i32),
    "sp_adam_step_flat": ([vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, i32, f32, vp], i32),
    "sp_adam_step_flat_dev": ([vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, vp, f32, vp], i32),
    "sp_adam_step_flat_hyp": ([vp, vp, vp, vp, i64, vp, vp, f32, vp], i32),
}
EXPORTS = sorted(list(_SIGS) + ["sp_last_error"])

_lib = None
_libs = {}
import threading as _threading
_tls = _threading.local()


def current_variant():
    return getattr(_tls, "variant", "")


class use:
    """``with use("f16"):`` -- calls made by this thread inside the block go to that build of the library (an engine is bound
    to one build: its tensors hold that build's 16-bit format)."""

    def __init__(self, variant):
        self.variant = variant or ""

    def __enter__(self):
        self.prev = current_variant()
        _tls.variant = self.variant
        return self

    def __exit__(self, *exc):
        _tls.variant = self.prev
        return False


def lib_path(variant=""):
    if variant == "":
        return LIB_PATH
    return os.path.join(os.path.dirname(LIB_PATH), VARIANTS[variant][0])


def load(variant=None):
    """Load the shared library (of the calling thread's current build, or the named one) once; raises if it has not been built."""
    global _lib
    variant = current_variant() if variant is None else variant
    if variant:
        if variant in _libs:
            return _libs[variant]
        path = lib_path(variant)
        import torch  # noqa: F401
        if not os.path.exists(path):
            raise RuntimeError("stroke_prediction_amd: %s is missing -- build it with `python -c \"import __graft_entry__ as g; "
                               "g.build()\"`. There is no CPU/PyTorch fallback." % path)
        lib = C.CDLL(path)
        for name, (argtypes, restype) in _SIGS.items():
            fn = getattr(lib, name)
            fn.argtypes = argtypes
            fn.restype = restype
        lib.sp_last_error.argtypes = [C.c_char_p, C.c_size_t]
        lib.sp_last_error.restype = None
        _libs[variant] = lib
        return lib
    if _lib is not None:
        return _lib
    # torch bundles its own HIP runtime (libamdhip64): import it FIRST so that this library binds to the
    # same runtime instance (streams and device pointers are shared with torch); loading ours first would
    # pull a second copy from /opt/rocm that never sees torch's context.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "stroke_prediction_amd: %s is missing -- build it with `python -c \"import __graft_entry__ as g; "
            "g.build()\"` (hipcc --offload-arch=gfx950). There is no CPU/PyTorch fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (argtypes, restype) in _SIGS.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = restype
    lib.sp_last_error.argtypes = [C.c_char_p, C.c_size_t]
    lib.sp_last_error.restype = None
    _lib = lib
    return lib


def last_error():
    buf = C.create_string_buffer(512)
    load().sp_last_error(buf, 512)
    return buf.value.decode(errors="replace")


def check(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed (rc=%d): %s" % (what, rc, last_error()))


def call(name, *args):
    """Call an ``int sp_*(...)`` entry point and raise on a non-zero return."""
    rc = getattr(load(), name)(*args)
    if rc != 0:
        raise RuntimeError("%s failed (rc=%d): %s" % (name, rc, last_error()))


def build(verbose=False):
    """Compile the HIP sources for gfx950 into ``lib/libstroke_amd.so`` and its precision variants (cross-compiles without a
    GPU).  One object per source and build (rebuilt only when stale, all compiled concurrently), then one link per build."""
    import subprocess
    os.makedirs(os.path.dirname(LIB_PATH), exist_ok=True)
    srcs = [os.path.join(CSRC_DIR, s) for s in SOURCES]
    hdrs = [os.path.join(CSRC_DIR, "sp_common.h"), os.path.join(os.path.dirname(PKG_DIR), "include", "stroke_amd.h")]
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    jobs, links = [], []
    for variant, (fname, flags) in VARIANTS.items():
        if os.environ.get("SP_LIB_PATH"):
            # a diagnostic build named by the environment (tools/build_variant*.sh, tools/build_asan.sh) is used as it is: nothing is
            # compiled beside it (the precision variants would land in ITS directory: six minutes of hipcc inside the sanitizer run)
            break
        out = lib_path(variant)
        if os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in srcs + hdrs):
            continue
        objdir = os.path.join(os.path.dirname(LIB_PATH), "obj" + ("_" + variant if variant else ""))
        os.makedirs(objdir, exist_ok=True)
        objs = []
        for src in srcs:
            obj = os.path.join(objdir, os.path.basename(src) + ".o")
            objs.append(obj)
            if not (os.path.exists(obj) and all(os.path.getmtime(obj) >= os.path.getmtime(d) for d in [src] + hdrs)):
                cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"] + flags + ["-c", src, "-o", obj]
                if verbose:
                    print(" ".join(cmd))
                jobs.append((cmd, subprocess.Popen(cmd)))
        links.append([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs + ["-ldl"])
    for cmd, pr in jobs:
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, cmd)
    for cmd in links:
        subprocess.run(cmd, check=True)
    global _lib
    if links:
        _lib = None
        _libs.clear()
    return LIB_PATH
