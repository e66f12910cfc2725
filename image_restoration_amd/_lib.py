"""ctypes binding of the C-ABI declared in ``include/mi_restore.h``.

The library is loaded lazily on first use and the load fails loudly: there is no fallback path.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MI_RESTORE_LIB") or os.path.join(HERE, "libmi_restore.so")   # (override: A/B builds)

MI_F32, MI_BF16 = 0, 1
c_i64 = C.c_int64
vp = C.c_void_p
fp = C.c_void_p  # float* passed as raw addresses


class PwDesc(C.Structure):
    _fields_ = [("x1", vp), ("x1_bs", c_i64), ("x1_gs", c_i64), ("k1", C.c_int),
                ("x2", vp), ("x2_bs", c_i64), ("x2_gs", c_i64), ("k2", C.c_int),
                ("w", fp), ("w_bs", c_i64), ("w_gs", c_i64), ("w_sm", c_i64), ("w_sk", c_i64),
                ("bias", fp), ("bias_gs", c_i64),
                ("r", vp), ("r_bs", c_i64), ("r_gs", c_i64),
                ("y", vp), ("y_bs", c_i64), ("y_gs", c_i64),
                ("m", C.c_int), ("n", c_i64), ("batch", C.c_int), ("groups", C.c_int), ("dtype", C.c_int),
                ("ln_w", fp), ("ln_b", fp), ("ln_mean", fp), ("ln_rstd", fp), ("ln_mode", C.c_int),
                ("f8", C.c_int), ("f8_sx", C.c_float), ("f8_sw", C.c_float),
                ("y_split", C.c_int), ("y2", vp), ("y2_bs", c_i64), ("y2_gs", c_i64),
                ("w_b16", vp), ("w_b16_sm", c_i64)]


class GramDesc(C.Structure):
    _fields_ = [("a", vp), ("a_bs", c_i64), ("a_gs", c_i64), ("ma", C.c_int),
                ("b", vp), ("b_bs", c_i64), ("b_gs", c_i64), ("mb", C.c_int),
                ("n", c_i64), ("batch", C.c_int), ("groups", C.c_int), ("dtype", C.c_int),
                ("sum_batch", C.c_int), ("accumulate", C.c_int),
                ("out", fp), ("out_ld", c_i64), ("out_zs", c_i64), ("sumsq", fp)]


class MdtaShape(C.Structure):
    _fields_ = [("B", C.c_int), ("C", C.c_int), ("heads", C.c_int), ("H", C.c_int), ("W", C.c_int),
                ("dtype", C.c_int), ("ks", C.c_int)]


class MdtaParams(C.Structure):
    _fields_ = [("temperature", fp), ("qkv_w", fp), ("qkv_b", fp), ("dw_w", fp), ("dw_b", fp),
                ("proj_w", fp), ("proj_b", fp)]


class MdtaGrads(C.Structure):
    _fields_ = [("temperature", fp), ("qkv_w", fp), ("qkv_b", fp), ("dw_w", fp), ("dw_b", fp),
                ("proj_w", fp), ("proj_b", fp), ("accumulate", C.c_int)]


class XmdtaShape(C.Structure):
    _fields_ = [("B", C.c_int), ("C", C.c_int), ("heads", C.c_int), ("H", C.c_int), ("W", C.c_int),
                ("dtype", C.c_int), ("ks_q", C.c_int), ("ks_kv", C.c_int)]


class XmdtaParams(C.Structure):
    _fields_ = [(n, fp) for n in ("temperature", "q_w", "q_b", "q_dw_w", "q_dw_b", "kv_w", "kv_b", "kv_dw_w", "kv_dw_b",
                                  "proj_w", "proj_b")]


class XmdtaGrads(C.Structure):
    _fields_ = [(n, fp) for n in ("temperature", "q_w", "q_b", "q_dw_w", "q_dw_b", "kv_w", "kv_b", "kv_dw_w", "kv_dw_b",
                                  "proj_w", "proj_b")] + [("accumulate", C.c_int)]


class GdfnShape(C.Structure):
    _fields_ = [("B", C.c_int), ("C", C.c_int), ("hidden", C.c_int), ("H", C.c_int), ("W", C.c_int),
                ("dtype", C.c_int), ("ks", C.c_int), ("flags", C.c_int)]


class GdfnParams(C.Structure):
    _fields_ = [("in_w", fp), ("in_b", fp), ("dw_w", fp), ("dw_b", fp), ("out_w", fp), ("out_b", fp)]


class GdfnFusedShape(C.Structure):
    _fields_ = [("B", C.c_int), ("C", C.c_int), ("hidden", C.c_int), ("H", C.c_int), ("W", C.c_int),
                ("ln_with_bias", C.c_int)]


class LnHead(C.Structure):
    _fields_ = [("w", fp), ("b", fp), ("mean", fp), ("rstd", fp), ("with_bias", C.c_int)]


class F8Scales(C.Structure):
    _fields_ = [("x1", C.c_float), ("w1", C.c_float), ("x2", C.c_float), ("w2", C.c_float)]


class LnTail(C.Structure):
    _fields_ = [("w", fp), ("b", fp), ("mean", fp), ("rstd", fp), ("dres", vp), ("dw", fp), ("db", fp)]


class GroupedProblem(C.Structure):
    _fields_ = [("x", vp), ("x_rs", c_i64), ("w", fp), ("w_sm", c_i64), ("w_sk", c_i64), ("bias", fp), ("r", vp), ("r_rs", c_i64),
                ("y", vp), ("y_rs", c_i64), ("m", C.c_int), ("k", C.c_int), ("expert", C.c_int), ("x_local", C.c_int),
                ("y_local", C.c_int), ("r_local", C.c_int)]


class GdfnGrads(C.Structure):
    _fields_ = [("in_w", fp), ("in_b", fp), ("dw_w", fp), ("dw_b", fp), ("out_w", fp), ("out_b", fp),
                ("accumulate", C.c_int)]


# symbol -> (restype, argtypes); this table is also what tests/test_cabi.py checks against the header
SIGNATURES = {
    "mi_version": (C.c_int, []),
    "mi_last_error": (C.c_char_p, []),
    "mi_env_reload": (C.c_int, []),
    "mi_deferred_begin": (C.c_int, [vp, C.c_size_t]),
    "mi_deferred_record": (C.c_int, [C.c_int]),
    "mi_deferred_pending": (C.c_int, []),
    "mi_deferred_high_water": (C.c_size_t, []),
    "mi_deferred_flush": (C.c_int, [vp]),
    "mi_deferred_end": (C.c_int, []),
    "mi_ln_fwd": (C.c_int, [vp, fp, fp, vp, fp, fp, C.c_int, C.c_int, c_i64, C.c_int, C.c_int, vp]),
    "mi_ln_bwd_workspace": (C.c_size_t, [C.c_int, C.c_int, c_i64]),
    "mi_ln_bwd": (C.c_int, [vp, vp, fp, fp, fp, vp, vp, fp, fp, C.c_int, C.c_int, c_i64, C.c_int, C.c_int, C.c_int,
                            vp, vp]),
    "mi_dwconv_fwd": (C.c_int, [vp, fp, fp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "mi_dwconv_gate_fwd": (C.c_int, [vp, fp, fp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "mi_dwconv_bwd_workspace": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "mi_dwconv_bwd": (C.c_int, [vp, vp, fp, vp, fp, fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                vp, vp]),
    "mi_dwconv_gate_bwd": (C.c_int, [vp, vp, vp, fp, vp, fp, fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.c_int, vp, vp]),
    "mi_dwconv_gate_recompute_ok": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "mi_dwconv_gate_bwd_recompute": (C.c_int, [vp, vp, fp, fp, vp, fp, fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                               C.c_int, C.c_int, vp, vp]),
    "mi_pw_gemm_workspace": (C.c_size_t, [C.POINTER(PwDesc)]),
    "mi_pw_gemm": (C.c_int, [C.POINTER(PwDesc), vp, vp]),
    "mi_pw_cache_enable": (C.c_int, [vp, C.c_size_t, vp, vp]),
    "mi_pw_cache_refresh": (C.c_int, [vp]),
    "mi_pw_cache_invalidate": (C.c_int, []),
    "mi_pw_cache_pending": (C.c_int, []),
    "mi_gram_workspace": (C.c_size_t, [C.POINTER(GramDesc)]),
    "mi_gram": (C.c_int, [C.POINTER(GramDesc), vp, vp]),
    "mi_mdta_saved_bytes": (C.c_size_t, [C.POINTER(MdtaShape)]),
    "mi_mdta_workspace": (C.c_size_t, [C.POINTER(MdtaShape)]),
    "mi_mdta_fwd": (C.c_int, [C.POINTER(MdtaShape), C.POINTER(MdtaParams), vp, vp, vp, vp, vp, vp]),
    "mi_mdta_bwd": (C.c_int, [C.POINTER(MdtaShape), C.POINTER(MdtaParams), vp, vp, vp, C.POINTER(MdtaGrads), vp, vp,
                              vp]),
    "mi_xmdta_saved_bytes": (C.c_size_t, [C.POINTER(XmdtaShape)]),
    "mi_xmdta_workspace": (C.c_size_t, [C.POINTER(XmdtaShape)]),
    "mi_xmdta_fwd": (C.c_int, [C.POINTER(XmdtaShape), C.POINTER(XmdtaParams), vp, vp, vp, vp, vp, vp, vp]),
    "mi_xmdta_bwd": (C.c_int, [C.POINTER(XmdtaShape), C.POINTER(XmdtaParams), vp, vp, vp, vp, vp, C.POINTER(XmdtaGrads), vp,
                               vp, vp]),
    "mi_gdfn_saved_bytes": (C.c_size_t, [C.POINTER(GdfnShape)]),
    "mi_gdfn_workspace": (C.c_size_t, [C.POINTER(GdfnShape)]),
    "mi_pw_gemm_ln_ok": (C.c_int, [C.POINTER(PwDesc)]),
    "mi_mdta_fwd_ln_ok": (C.c_int, [C.POINTER(MdtaShape)]),
    "mi_mdta_fwd_ln": (C.c_int, [C.POINTER(MdtaShape), C.POINTER(MdtaParams), C.POINTER(LnHead), vp, vp, vp, vp, vp, vp]),
    "mi_gdfn_fwd_ln_ok": (C.c_int, [C.POINTER(GdfnShape)]),
    "mi_gdfn_fwd_ln": (C.c_int, [C.POINTER(GdfnShape), C.POINTER(GdfnParams), C.POINTER(LnHead), vp, vp, vp, vp, vp, vp]),
    "mi_pw_gemm_f8_ok": (C.c_int, [C.POINTER(PwDesc)]),
    "mi_pw_gemm_split_ok": (C.c_int, [C.POINTER(PwDesc)]),
    "mi_mdta_fwd_f8_ok": (C.c_int, [C.POINTER(MdtaShape), C.c_int]),
    "mi_mdta_fwd_f8": (C.c_int, [C.POINTER(MdtaShape), C.POINTER(MdtaParams), C.POINTER(LnHead), C.POINTER(F8Scales), vp, vp, vp,
                                 vp, vp]),
    "mi_gdfn_fwd_f8_ok": (C.c_int, [C.POINTER(GdfnShape), C.c_int]),
    "mi_gdfn_fwd_f8": (C.c_int, [C.POINTER(GdfnShape), C.POINTER(GdfnParams), C.POINTER(LnHead), C.POINTER(F8Scales), vp, vp, vp,
                                 vp, vp]),
    "mi_box_down": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "mi_fre_rect": (C.c_int, [fp, fp, fp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "mi_fre_split_coef_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "mi_fre_split_workspace": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "mi_fre_split_max_hw": (C.c_int, []),
    "mi_fre_split_fwd": (C.c_int, [vp, vp, vp, vp, fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "mi_fre_split_bwd": (C.c_int, [vp, vp, fp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]),
    "mi_chan_maxmean_fwd": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, c_i64, C.c_int, vp]),
    "mi_chan_maxmean_bwd": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, c_i64, C.c_int, vp]),
    "mi_plane_max_fwd": (C.c_int, [vp, fp, vp, C.c_int, c_i64, C.c_int, vp]),
    "mi_pool_pair_bwd": (C.c_int, [fp, fp, vp, vp, C.c_int, c_i64, C.c_int, vp]),
    "mi_chan_gate_fwd": (C.c_int, [fp, fp, fp, fp, fp, fp, C.c_int, C.c_int, C.c_int, vp]),
    "mi_chan_gate_bwd": (C.c_int, [fp, fp, fp, fp, fp, fp, fp, fp, fp, fp, fp, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "mi_refine_mix_fwd": (C.c_int, [vp, vp, vp, fp, vp, C.c_int, C.c_int, c_i64, C.c_int, vp]),
    "mi_refine_mix_bwd": (C.c_int, [vp, vp, vp, fp, vp, vp, vp, vp, fp, C.c_int, C.c_int, c_i64, C.c_int, vp]),
    "mi_scale_add_fwd": (C.c_int, [vp, vp, fp, fp, vp, C.c_int, C.c_int, c_i64, C.c_int, vp]),
    "mi_scale_add_bwd": (C.c_int, [vp, vp, fp, fp, vp, vp, vp, fp, fp, C.c_int, C.c_int, c_i64, C.c_int, C.c_int, vp]),
    "mi_bwd_tail_ok": (C.c_int, [C.c_int, C.c_int, c_i64, C.c_int]),
    "mi_bwd_tail_workspace": (C.c_size_t, [C.c_int, C.c_int]),
    "mi_bwd_tail": (C.c_int, [vp, C.c_int, vp, C.c_int, vp, fp, fp, fp, fp, fp, vp, fp, fp, fp, C.c_int, c_i64, C.c_int,
                              C.c_int, vp, vp]),
    "mi_mdta_bwd_ln_ok": (C.c_int, [C.POINTER(MdtaShape), C.c_int]),
    "mi_mdta_bwd_ln_workspace": (C.c_size_t, [C.POINTER(MdtaShape)]),
    "mi_mdta_bwd_ln": (C.c_int, [C.POINTER(MdtaShape), C.POINTER(MdtaParams), C.POINTER(LnTail), vp, vp, vp,
                                 C.POINTER(MdtaGrads), vp, vp, vp]),
    "mi_gdfn_bwd_ln_ok": (C.c_int, [C.POINTER(GdfnShape), C.c_int]),
    "mi_gdfn_bwd_ln_workspace": (C.c_size_t, [C.POINTER(GdfnShape)]),
    "mi_gdfn_bwd_ln": (C.c_int, [C.POINTER(GdfnShape), C.POINTER(GdfnParams), C.POINTER(LnTail), vp, vp, vp,
                                 C.POINTER(GdfnGrads), vp, vp, vp]),
    "mi_gdfn_fwd": (C.c_int, [C.POINTER(GdfnShape), C.POINTER(GdfnParams), vp, vp, vp, vp, vp, vp]),
    "mi_gdfn_bwd": (C.c_int, [C.POINTER(GdfnShape), C.POINTER(GdfnParams), vp, vp, vp, C.POINTER(GdfnGrads), vp, vp,
                              vp]),
    "mi_gdfn_fused_ok": (C.c_int, [C.POINTER(GdfnFusedShape)]),
    "mi_gdfn_fused_pack_bytes": (C.c_size_t, [C.POINTER(GdfnFusedShape)]),
    "mi_gdfn_fused_pack": (C.c_int, [C.POINTER(GdfnFusedShape), fp, fp, C.POINTER(GdfnParams), vp, vp]),
    "mi_gdfn_fused_fwd": (C.c_int, [C.POINTER(GdfnFusedShape), vp, vp, vp, fp, fp, vp]),
    "mi_gdfn_fused_fwd_train_ok": (C.c_int, [C.POINTER(GdfnFusedShape)]),
    "mi_gdfn_fused_fwd_train": (C.c_int, [C.POINTER(GdfnFusedShape), vp, vp, vp, fp, fp, vp, vp]),
    "mi_gdfn_fused_fwd_f8": (C.c_int, [C.POINTER(GdfnFusedShape), vp, C.POINTER(F8Scales), vp, vp, vp]),
    "mi_mdta_fused_ok": (C.c_int, [C.POINTER(MdtaShape)]),
    "mi_mdta_fused_pays": (C.c_int, [C.POINTER(MdtaShape)]),
    "mi_mdta_fused_pack_bytes": (C.c_size_t, [C.POINTER(MdtaShape)]),
    "mi_mdta_fused_pack": (C.c_int, [C.POINTER(MdtaShape), fp, fp, C.POINTER(MdtaParams), vp, vp]),
    "mi_mdta_fused_workspace": (C.c_size_t, [C.POINTER(MdtaShape)]),
    "mi_mdta_fused_fwd": (C.c_int, [C.POINTER(MdtaShape), C.POINTER(MdtaParams), vp, C.c_int, vp, vp, vp, fp, fp, vp, vp]),
    "mi_adamw_step": (C.c_int, [fp, fp, fp, fp, c_i64, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int,
                                C.c_float, fp, vp]),
    "mi_rows_gather": (C.c_int, [vp, vp, vp, C.c_int, c_i64, C.c_int, vp]),
    "mi_rows_gather_scaled": (C.c_int, [fp, vp, fp, vp, C.c_int, c_i64, C.c_int, vp]),
    "mi_rows_scatter_add": (C.c_int, [vp, vp, fp, vp, C.c_int, C.c_int, c_i64, C.c_int, C.c_int, vp]),
    "mi_rows_dot_workspace": (C.c_size_t, [C.c_int, c_i64]),
    "mi_rows_dot": (C.c_int, [fp, vp, vp, fp, C.c_int, c_i64, C.c_int, vp, vp]),
    "mi_glue3x3_ok": (C.c_int, [C.c_int, C.c_int]),
    "mi_conv3x3_ok": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "mi_conv3x3_pack_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "mi_conv3x3_pack": (C.c_int, [fp, C.c_int, C.c_int, C.c_int, vp, vp]),
    "mi_conv3x3_fwd": (C.c_int, [vp, vp, C.c_int64, fp, vp, C.c_int64, vp, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "mi_conv3x3_wgrad_workspace": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "mi_conv3x3_wgrad": (C.c_int, [vp, C.c_int64, vp, C.c_int64, fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]),
    "mi_chan_sum_workspace": (C.c_size_t, [C.c_int, C.c_int64]),
    "mi_chan_sum": (C.c_int, [vp, fp, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_int, vp, vp]),
    "mi_im2col3x3": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "mi_col2im3x3": (C.c_int, [vp, fp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "mi_grouped_pw_gemm": (C.c_int, [C.POINTER(GroupedProblem), C.c_int, vp, vp, C.c_int, c_i64, C.c_int, vp]),
    "mi_moe_route_fwd": (C.c_int, [fp, fp, fp, fp, fp, fp, fp, fp, vp, fp, fp, vp, vp, vp, fp, vp, vp, C.c_int, C.c_int, C.c_int,
                                   C.c_int, C.c_int, C.c_int, vp]),
    "mi_moe_route_bwd": (C.c_int, [fp, fp, fp, fp, fp, fp, fp, vp, fp, fp, vp, fp, fp, fp, fp, fp, C.c_int, C.c_int, C.c_int,
                                   C.c_int, C.c_int, C.c_int, vp]),
    "mi_patch_circconv": (C.c_int, [vp, c_i64, vp, c_i64, vp, c_i64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "mi_gelu_gap_fwd": (C.c_int, [vp, fp, C.c_int, C.c_int, c_i64, C.c_int, vp]),
    "mi_gelu_gap_bwd": (C.c_int, [vp, fp, vp, C.c_int, C.c_int, c_i64, C.c_int, vp]),
    "mi_ewise_fwd": (C.c_int, [vp, c_i64, vp, c_i64, vp, c_i64, c_i64, C.c_int, C.c_int, vp]),
    "mi_ewise_bwd": (C.c_int, [vp, c_i64, vp, c_i64, vp, vp, c_i64, vp, c_i64, c_i64, c_i64, C.c_int, C.c_int, vp]),
    "mi_pixel_shuffle2": (C.c_int, [vp, c_i64, vp, c_i64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "mi_copy_rows": (C.c_int, [vp, c_i64, vp, c_i64, c_i64, c_i64, C.c_int, vp]),
    "mi_patch_batch": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, fp, fp, vp, vp, C.c_int, C.c_int, C.c_int, vp]),
    "mi_psnr_ssim_workspace": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "mi_psnr_ssim": (C.c_int, [vp, vp, fp, fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]),
    "mi_gap_fwd": (C.c_int, [vp, fp, C.c_int, C.c_int, c_i64, C.c_int, vp]),
    "mi_gap_bwd": (C.c_int, [fp, vp, C.c_int, C.c_int, c_i64, C.c_int, vp]),
    "mi_cast": (C.c_int, [vp, C.c_int, vp, C.c_int, c_i64, vp]),
    "mi_l1_loss": (C.c_int, [vp, vp, vp, fp, c_i64, C.c_float, C.c_int, vp]),
    "mi_prof_enable": (C.c_int, [C.c_int]),
    "mi_prof_kernel_count": (C.c_int, []),
    "mi_prof_kernel_name": (C.c_char_p, [C.c_int]),
    "mi_prof_collect": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                  C.POINTER(c_i64), C.c_int]),
}

_lib: Optional[C.CDLL] = None


def lib() -> C.CDLL:
    """The loaded library; raises RuntimeError (never falls back) when it is missing or incomplete."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(hipcc --offload-arch=gfx950); there is no CPU fallback")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the symbol is missing
            fn.restype, fn.argtypes = res, args
        _lib = handle
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().mi_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"mi_restore {what} failed (rc={rc}): {msg}")
