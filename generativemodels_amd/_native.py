"""ctypes binding of the C-ABI kernel library (include/gm_amd.h -> generativemodels_amd/lib/libgmamd.so).

There is no CPU or eager-PyTorch fallback behind this module: if the library is missing, or a tensor is not resident on an
MI355X, every op raises. torch is used for device memory and streams only."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# GM_NATIVE_LIB: a bench-only build of the same library (e.g. lib/libgmamd_timeline.so from `_build.py --variant timeline`); never set in production
LIB_PATH = os.environ.get("GM_NATIVE_LIB") or os.path.join(_HERE, "lib", "libgmamd.so")

c_ll = C.c_longlong
c_vp = C.c_void_p


class GmStepParams(C.Structure):
    _fields_ = [("mode", C.c_int), ("pred_type", C.c_int), ("c_sa", C.c_float), ("c_sb", C.c_float), ("clip", C.c_int),
                ("clip_lo", C.c_float), ("clip_hi", C.c_float), ("c_prev", C.c_float), ("c_dir", C.c_float),
                ("k0", C.c_float), ("k1", C.c_float), ("noise_mode", C.c_int), ("c_noise", C.c_float),
                ("min_log", C.c_float), ("max_log", C.c_float)]


class GmGnTables(C.Structure):
    _fields_ = [("stats", C.c_void_p * 2), ("S", C.c_int * 2), ("C", C.c_int * 2), ("gamma", C.c_void_p), ("beta", C.c_void_p), ("eps", C.c_float),
                ("groups", C.c_int)]


class GmKlParams(C.Structure):
    _fields_ = [("pred_type", C.c_int), ("c_sa", C.c_float), ("c_sb", C.c_float), ("clip", C.c_int), ("k0", C.c_float),
                ("k1", C.c_float), ("m0", C.c_float), ("m1", C.c_float), ("t0", C.c_int), ("s", C.c_float), ("e", C.c_float),
                ("half_bin", C.c_float)]


class GmConvDesc(C.Structure):
    _fields_ = [("x", c_vp), ("x_ld", c_ll), ("w", c_vp), ("bias", c_vp), ("pre_scale", c_vp), ("pre_shift", c_vp),
                ("rowvec", c_vp), ("rowvec_bstride", c_ll), ("res", c_vp), ("res_ld", c_ll), ("y", c_vp), ("y_ld", c_ll),
                ("N", C.c_int), ("Cin", C.c_int), ("Cout", C.c_int),
                ("Ds", C.c_int), ("Hs", C.c_int), ("Ws", C.c_int), ("Do", C.c_int), ("Ho", C.c_int), ("Wo", C.c_int),
                ("kd", C.c_int), ("kh", C.c_int), ("kw", C.c_int), ("sd", C.c_int), ("sh", C.c_int), ("sw", C.c_int),
                ("pd", C.c_int), ("ph", C.c_int), ("pw", C.c_int), ("dd", C.c_int), ("dh", C.c_int), ("dw", C.c_int),
                ("in_mode", C.c_int), ("fd", C.c_int), ("fh", C.c_int), ("fw", C.c_int),
                ("pre_act", C.c_int), ("post_act", C.c_int), ("dtype", C.c_int),
                ("ltd", C.c_int), ("lth", C.c_int), ("ltw", C.c_int), ("cfg", C.c_int), ("debug_flags", C.c_int), ("stats", c_vp),
                ("skip_x", c_vp * 2), ("skip_ld", c_ll * 2), ("skip_cin", C.c_int * 2), ("skip_w", c_vp), ("skip_bias", c_vp),
                ("x2", c_vp), ("x2_ld", c_ll), ("cin_split", C.c_int), ("ksplit", C.c_int), ("kpartial", c_vp),
                ("pre_stats", c_vp * 2), ("pre_S", C.c_int * 2), ("pre_C", C.c_int * 2), ("pre_gamma", c_vp), ("pre_beta", c_vp), ("pre_eps", C.c_float),
                ("pre_groups", C.c_int)]


class GmDecodeBlock(C.Structure):
    _fields_ = [("ln1_g", c_vp), ("ln1_b", c_vp), ("w_qkv", c_vp), ("b_qkv", c_vp), ("w_o", c_vp), ("b_o", c_vp), ("ln3_g", c_vp),
                ("ln3_b", c_vp), ("w_1", c_vp), ("b_1", c_vp), ("w_2", c_vp), ("b_2", c_vp), ("k_cache", c_vp), ("v_cache", c_vp)]


class GmDecodeDesc(C.Structure):
    _fields_ = [("B", C.c_int), ("C", C.c_int), ("M", C.c_int), ("heads", C.c_int), ("depth", C.c_int), ("max_len", C.c_int),
                ("num_tokens", C.c_int), ("dtype", C.c_int), ("ln_eps", C.c_float), ("pos", C.c_int), ("tokens", c_vp),
                ("tok_emb", c_vp), ("pos_emb", c_vp), ("blocks", C.POINTER(GmDecodeBlock)), ("w_logits", c_vp), ("b_logits", c_vp),
                ("logits", c_vp), ("scratch", c_vp), ("scratch_bytes", c_ll), ("pos_dev", c_vp)]


class GmAttnDesc(C.Structure):
    _fields_ = [("q", c_vp), ("q_ld", c_ll), ("k", c_vp), ("k_ld", c_ll), ("v", c_vp), ("v_ld", c_ll),
                ("res", c_vp), ("res_ld", c_ll), ("o", c_vp), ("o_ld", c_ll),
                ("B", C.c_int), ("H", C.c_int), ("Lq", C.c_int), ("Lk", C.c_int), ("dh", C.c_int),
                ("scale", C.c_float), ("dtype", C.c_int), ("workspace", c_vp), ("workspace_bytes", c_ll),
                ("causal", C.c_int), ("k_bs", c_ll), ("v_bs", c_ll), ("stats", c_vp), ("vt_packed", C.c_int), ("lse", c_vp)]


class GmAttnBwdDesc(C.Structure):
    _fields_ = [("q", c_vp), ("q_ld", c_ll), ("k", c_vp), ("k_ld", c_ll), ("v", c_vp), ("v_ld", c_ll), ("o", c_vp), ("o_ld", c_ll),
                ("go", c_vp), ("go_ld", c_ll), ("dq", c_vp), ("dq_ld", c_ll), ("dk", c_vp), ("dk_ld", c_ll), ("dv", c_vp), ("dv_ld", c_ll),
                ("B", C.c_int), ("H", C.c_int), ("Lq", C.c_int), ("Lk", C.c_int), ("dh", C.c_int), ("scale", C.c_float), ("dtype", C.c_int),
                ("workspace", c_vp), ("workspace_bytes", c_ll)]


class GmWgradDesc(C.Structure):
    _fields_ = [("x", c_vp), ("x_ld", c_ll), ("gy", c_vp), ("gy_ld", c_ll), ("dw", c_vp), ("workspace", c_vp), ("workspace_bytes", c_ll),
                ("N", C.c_int), ("Cin", C.c_int), ("Cout", C.c_int), ("Ds", C.c_int), ("Hs", C.c_int), ("Ws", C.c_int),
                ("Do", C.c_int), ("Ho", C.c_int), ("Wo", C.c_int), ("kd", C.c_int), ("kh", C.c_int), ("kw", C.c_int),
                ("stride", C.c_int), ("pd", C.c_int), ("ph", C.c_int), ("pw", C.c_int), ("dtype", C.c_int), ("accumulate", C.c_int)]


# name -> (restype, argtypes); mirrors include/gm_amd.h one to one (tests/test_abi.py checks the export list)
PROTOTYPES = {
    "gm_abi_version": (C.c_int, []),
    "gm_last_error": (C.c_char_p, []),
    "gm_sched_step": (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_ll, c_ll, c_ll, C.c_int, C.POINTER(GmStepParams), c_vp]),
    "gm_nearest_resize": (C.c_int, [c_vp, c_ll, c_vp, c_ll] + [C.c_int] * 9 + [c_vp]),
    "gm_axpby_rows": (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_ll, c_ll, C.c_int, c_vp]),
    "gm_likelihood_term": (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_ll, c_ll, c_ll, C.c_int, C.POINTER(GmKlParams), c_vp]),
    "gm_lincomb": (C.c_int, [C.POINTER(c_vp), C.POINTER(C.c_float), C.c_int, C.c_float, C.c_float, c_vp, c_ll, C.c_int, c_vp]),
    "gm_copy_channels": (C.c_int, [c_vp, c_ll, C.c_int, c_vp, c_ll, C.c_int, c_ll, C.c_int, c_vp]),
    "gm_normal_bf16_from_bits": (C.c_int, [c_vp, c_vp, c_vp, c_ll, c_vp]),
    "gm_sched_step_noise_bits": (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_ll, c_ll, c_ll, C.POINTER(GmStepParams), c_vp]),
    "gm_token_gemm_set_wide": (None, [C.c_int, C.c_int]),
    "gm_attention_set_wave_groups": (None, [C.c_int]),
    "gm_nchw_to_nhwc": (C.c_int, [c_vp, C.c_int, c_vp, C.c_int, C.c_int, C.c_int, c_ll, c_ll, c_vp]),
    "gm_nhwc_to_nchw": (C.c_int, [c_vp, c_ll, C.c_int, c_vp, C.c_int, C.c_int, C.c_int, c_ll, c_vp]),
    "gm_resample2x": (C.c_int, [c_vp, c_ll, c_vp, c_ll, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                C.c_int, c_vp]),
    "gm_phase2x": (C.c_int, [c_vp, c_ll, c_vp, c_ll, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_vp]),
    "gm_timestep_embedding": (C.c_int, [c_vp, c_vp, C.c_int, C.c_int, C.c_float, C.c_int, c_vp]),
    "gm_geglu": (C.c_int, [c_vp, c_ll, c_vp, c_ll, c_ll, C.c_int, C.c_int, c_vp]),
    "gm_aekl_sample": (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_ll, C.c_int, c_vp]),
    "gm_addcmul": (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_ll, C.c_int, c_vp]),
    "gm_scale": (C.c_int, [c_vp, c_vp, C.c_float, C.c_int, c_ll, C.c_int, c_vp]),
    "gm_activation": (C.c_int, [c_vp, c_vp, c_vp, C.c_int, C.c_int, c_ll, C.c_int, c_vp]),
    "gm_gn_workspace_bytes": (c_ll, [C.c_int, c_ll, C.c_int, C.c_int, C.c_int]),
    "gm_gn_scale_shift": (C.c_int, [c_vp, c_ll, C.c_int, c_ll, C.c_int, C.c_int, C.c_float, c_vp, c_vp, c_vp, c_vp, c_vp,
                                    c_vp, c_vp, C.c_int, c_vp]),
    "gm_gn_apply": (C.c_int, [c_vp, c_ll, c_vp, c_ll, c_vp, c_vp, c_ll, C.c_int, c_ll, C.c_int, C.c_int, C.c_int, c_vp]),
    "gm_spade_apply": (C.c_int, [c_vp, c_ll, c_vp, c_ll, c_vp, c_vp, c_ll, c_vp, c_vp, c_ll, C.c_int, c_ll, C.c_int, C.c_int, C.c_int, c_vp]),
    "gm_gn_channel_stats": (C.c_int, [c_vp, c_ll, C.c_int, c_ll, C.c_int, c_vp, C.c_int, c_vp]),
    "gm_gn_channel_stats_slots": (c_ll, [c_vp, c_ll, c_ll, C.c_int, C.c_int]),
    "gm_stats_compact_slots": (C.c_int, []),
    "gm_stats_compact": (C.c_int, [c_vp, C.c_int, C.c_int, C.c_int, c_vp, c_vp]),
    "gm_gn_finalize_channels": (C.c_int, [c_vp, C.c_int, C.c_int, c_vp, C.c_int, C.c_int, C.c_int, c_ll, C.c_int, C.c_float, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "gm_layernorm": (C.c_int, [c_vp, c_ll, c_vp, c_ll, c_vp, c_vp, c_ll, C.c_int, C.c_float, C.c_int, c_vp]),
    "gm_conv_cfg_tile": (C.c_int, [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "gm_conv_lds_bytes": (c_ll, [C.POINTER(GmConvDesc)]),
    "gm_conv_stats_slots": (c_ll, [C.POINTER(GmConvDesc)]),
    "gm_conv_splitk_workspace_bytes": (c_ll, [C.POINTER(GmConvDesc)]),
    "gm_conv_forward": (C.c_int, [C.POINTER(GmConvDesc), c_vp]),
    "gm_conv_dma_set_persistent": (None, [C.c_int]),
    "gm_conv_dma_set_phase_skew": (None, [C.c_int]),
    "gm_conv_sk_set_enabled": (None, [C.c_int]),
    "gm_packed_conv_weight_elems": (c_ll, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "gm_pack_conv_weight": (C.c_int, [c_vp, C.c_int, c_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                      c_vp]),
    "gm_pack_subpixel_weight": (C.c_int, [c_vp, C.c_int, c_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_vp]),
    "gm_attention_max_head_dim": (C.c_int, []),
    "gm_linear_rows": (C.c_int, [c_vp, c_ll, c_vp, c_vp, c_vp, c_ll, c_vp, c_ll, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_vp]),
    "gm_linear_rows_affine_vt": (C.c_int, [c_vp, c_ll, c_vp, c_vp, c_ll, C.c_int, c_vp, c_vp, c_vp, c_ll, C.c_int, C.c_int, C.c_int, C.c_int, c_vp, C.c_int,
                                           C.c_int, C.c_int, c_vp]),
    "gm_linear_rows_gn": (C.c_int, [c_vp, c_ll, C.POINTER(GmGnTables), C.c_int, C.c_int, c_vp, c_vp, c_vp, c_ll, c_vp, c_ll, C.c_int, C.c_int, C.c_int, C.c_int,
                                    C.c_int, c_vp, C.c_int, C.c_int, C.c_int, c_vp]),
    "gm_linear_rows_affine": (C.c_int, [c_vp, c_ll, c_vp, c_vp, c_ll, C.c_int, c_vp, c_vp, c_vp, c_ll, c_vp, c_ll, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.c_int, C.c_int, c_vp]),
    "gm_decode_scratch_bytes": (c_ll, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "gm_decode_advance": (C.c_int, [c_vp, c_vp, c_vp, c_vp, C.c_int, c_ll, c_vp]),
    "gm_transformer_decode_step": (C.c_int, [C.POINTER(GmDecodeDesc), c_vp]),
    "gm_embed_tokens": (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_ll, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_vp]),
    "gm_sample_probs": (C.c_int, [c_vp, c_ll, c_vp, c_ll, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, c_vp]),
    "gm_sample_index": (C.c_int, [c_vp, c_ll, C.c_int, c_vp, c_vp, c_vp]),
    "gm_token_log_prob": (C.c_int, [c_vp, c_ll, c_vp, c_vp, c_ll, C.c_int, C.c_int, c_vp]),
    "gm_attention_workspace_bytes": (c_ll, [C.POINTER(GmAttnDesc)]),
    "gm_attention_stats_slots": (c_ll, [C.POINTER(GmAttnDesc)]),
    "gm_attention_dma_set_variant": (None, [C.c_int, C.c_int]),
    "gm_attention_forward": (C.c_int, [C.POINTER(GmAttnDesc), c_vp]),
    "gm_conv_wgrad_workspace_bytes": (c_ll, [C.POINTER(GmWgradDesc)]),
    "gm_conv_wgrad": (C.c_int, [C.POINTER(GmWgradDesc), c_vp]),
    "gm_gn_bwd_stats": (C.c_int, [c_vp, c_ll, c_vp, c_ll, c_vp, c_vp, c_ll, C.c_int, c_ll, C.c_int, C.c_int, c_vp, C.c_int, c_vp]),
    "gm_gn_bwd_stats_slots": (c_ll, [C.c_int, c_ll]),
    "gm_gn_bwd_finalize": (C.c_int, [c_vp, C.c_int, c_vp, C.c_int, C.c_int, C.c_int, C.c_int, c_ll, C.c_float, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "gm_layernorm_bwd_slots": (C.c_int, [c_ll]),
    "gm_likelihood_workspace_elems": (c_ll, [c_ll, c_ll]),
    "gm_gn_bwd_apply": (C.c_int, [c_vp, c_ll, c_vp, c_ll, c_vp, c_ll, c_vp, c_vp, c_ll, c_vp, c_vp, c_vp, C.c_int, c_ll, C.c_int, C.c_int,
                                  C.c_int, c_vp]),
    "gm_spade_bwd": (C.c_int, [c_vp, c_ll, c_vp, c_vp, c_ll, c_vp, c_ll, c_vp, c_ll, c_vp, c_vp, c_ll, c_ll, C.c_int, C.c_int, C.c_int, c_vp]),
    "gm_stats_colsum": (C.c_int, [c_vp, C.c_int, C.c_int, C.c_int, c_vp, C.c_int, c_vp]),
    "gm_attention_backward_workspace_bytes": (c_ll, [C.POINTER(GmAttnBwdDesc)]),
    "gm_attention_backward": (C.c_int, [C.POINTER(GmAttnBwdDesc), c_vp]),
    "gm_attention_backward_fused_workspace_bytes": (c_ll, [C.POINTER(GmAttnBwdDesc)]),
    "gm_attention_backward_fused": (C.c_int, [C.POINTER(GmAttnBwdDesc), c_vp, c_vp]),
    "gm_attention_backward_fused_set_split": (None, [C.c_int]),
    "gm_attention_bwd_scores_workspace_bytes": (c_ll, [C.POINTER(GmAttnBwdDesc)]),
    "gm_attention_bwd_scores": (C.c_int, [C.POINTER(GmAttnBwdDesc), c_vp, c_vp, c_ll, c_vp, c_ll, c_vp]),
    "gm_layernorm_bwd": (C.c_int, [c_vp, c_ll, c_vp, c_ll, c_vp, c_ll, c_vp, c_ll, C.c_int, C.c_float, c_vp, C.c_int, c_vp]),
    "gm_geglu_bwd": (C.c_int, [c_vp, c_ll, c_vp, c_ll, c_vp, c_ll, c_ll, C.c_int, C.c_int, c_vp]),
    "gm_softmax_bwd": (C.c_int, [c_vp, c_vp, c_vp, c_ll, C.c_int, C.c_float, c_vp]),
    "gm_vq_argmin": (C.c_int, [c_vp, c_ll, c_vp, c_vp, c_ll, C.c_int, C.c_int, C.c_int, c_vp]),
    "gm_vq_gather_workspace_bytes": (c_ll, []),
    "gm_vq_ema_stats_workspace_elems": (c_ll, [c_ll, C.c_int, C.c_int]),
    "gm_vq_ema_stats": (C.c_int, [c_vp, c_ll, c_vp, c_ll, C.c_int, C.c_int, c_vp, c_vp, C.c_int, c_vp]),
    "gm_vq_ema_update": (C.c_int, [c_vp, c_vp, c_vp, c_vp, C.c_int, C.c_int, C.c_float, C.c_float, c_vp]),
    "gm_vq_gather": (C.c_int, [c_vp, c_vp, c_vp, c_ll, c_vp, c_ll, c_vp, c_vp, c_ll, C.c_int, C.c_int, C.c_int, c_vp]),
}

_lib = None


class NativeLibraryError(RuntimeError):
    pass


def lib() -> C.CDLL:
    """The loaded C-ABI library. Raises NativeLibraryError (never falls back) when it has not been built."""
    global _lib
    if _lib is None:
        import torch  # noqa: F401  loads the HIP runtime the library shares with the host framework (see _build.hip_runtime_library)

        if not os.path.exists(LIB_PATH):
            raise NativeLibraryError(
                f"{LIB_PATH} is missing: build it with `python -m generativemodels_amd._build` (hipcc, gfx950). "
                "generativemodels_amd has no CPU / eager fallback.")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(handle, name)  # AttributeError here = header / library mismatch
            fn.restype = res
            fn.argtypes = args
        if os.environ.get("GM_CONV_DMA_GRID"):  # A/B of the LDS-DMA grid policy (gm_conv_dma_set_persistent): -1 auto, 0 one tile per work-group
            handle.gm_conv_dma_set_persistent(int(os.environ["GM_CONV_DMA_GRID"]))
        if os.environ.get("GM_CONV_DMA_SKEW"):  # A/B of the one-time phase offset (gm_conv_dma_set_phase_skew): cycles, 0 off, -1 automatic
            handle.gm_conv_dma_set_phase_skew(int(os.environ["GM_CONV_DMA_SKEW"]))
        if os.environ.get("GM_ATTN_WAVE_GROUPS"):  # A/B of the register-staged attention kernel's wave groups (gm_attention_set_wave_groups)
            handle.gm_attention_set_wave_groups(int(os.environ["GM_ATTN_WAVE_GROUPS"]))
        if os.environ.get("GM_TOKEN_GEMM_WIDE") or os.environ.get("GM_TOKEN_GEMM_WIDE_NB"):  # A/B of the wide token GEMM: "0" off, else the row threshold; blocks per wave
            rows = int(os.environ.get("GM_TOKEN_GEMM_WIDE", "-1"))
            handle.gm_token_gemm_set_wide(rows, int(os.environ.get("GM_TOKEN_GEMM_WIDE_NB", "0")))
        _lib = handle
    return _lib


_DEBUG_SYNC = bool(os.environ.get("GM_DEBUG_SYNC"))  # synchronise after every native call and name the one that faulted


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().gm_last_error()
        raise RuntimeError(f"libgmamd {what} failed (code {rc}): {msg.decode() if msg else '?'}")
    if _DEBUG_SYNC:
        import torch

        if not torch.cuda.is_current_stream_capturing():
            print(f"[gm] {what}", flush=True)
            torch.cuda.synchronize()
