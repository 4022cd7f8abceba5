"""ctypes binding of libadvgrpo_hip.so (the C ABI declared in include/advgrpo.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C adv_grpo_amd/csrc``.
There is NO CPU fallback: if the shared object is missing or a symbol cannot be resolved the
import of any product module that needs a kernel raises, loudly.
"""
import ctypes
import os
from ctypes import c_double, c_float, c_int, c_int32, c_int64, c_uint64, c_void_p, POINTER

import torch

F32, BF16, F64 = 0, 1, 2
SDE_EPS, SDE_PHILOX, SDE_REPLAY = 0, 1, 2

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ADVGRPO_LIB") or os.path.join(_HERE, "libadvgrpo_hip.so")   # override: A/B runs of two builds

_P = c_void_p


class GemmDesc(ctypes.Structure):
    """advgrpo_gemm_desc (include/advgrpo.h), field for field."""
    _fields_ = [("A", _P), ("W", _P), ("C", _P), ("lda", c_int64), ("ldw", c_int64), ("ldc", c_int64),
                ("out_dtype", c_int32), ("M", c_int32), ("N", c_int32), ("K", c_int32),
                ("bias", _P), ("act", c_int32), ("alpha", c_float),
                ("gate", _P), ("gate_stride", c_int64), ("gate_rows", c_int32),
                ("residual", _P), ("ldr", c_int64),
                ("seg_rows", c_int32), ("seg_stride", c_int64), ("seg_off", c_int64),
                ("a_seg_rows", c_int32), ("a_seg_stride", c_int64), ("a_seg_off", c_int64),
                ("aux_out", _P), ("aux_in", _P), ("ld_aux", c_int64),
                ("rms_weight", _P), ("rms_nheads", c_int32), ("rms_heads_per_weight", c_int32), ("rms_eps", c_float),
                ("rms_rs_out", _P)]


class LnDesc(ctypes.Structure):
    """advgrpo_ln_desc (include/advgrpo.h), field for field."""
    _fields_ = [("x", _P), ("ldx", c_int64), ("out0", _P), ("out1", _P), ("ldo", c_int64), ("w", _P), ("b", _P),
                ("scale0", _P), ("shift0", _P), ("scale1", _P), ("shift1", _P), ("mod_stride", c_int64),
                ("rows_per_batch", c_int32), ("M", c_int32), ("D", c_int32), ("eps", c_float),
                ("q0", _P), ("qs0", _P), ("q1", _P), ("qs1", _P), ("ldq", c_int64)]


class MMDiTBlockDesc(ctypes.Structure):
    """advgrpo_mmdit_block_desc (include/advgrpo.h), field for field."""
    _fields_ = ([(n, c_int32) for n in ("B", "Ni", "Nt", "D", "H", "dual", "last")] + [("x", _P), ("c", _P), ("mods", _P)] +
                [(n, c_int64) for n in ("mod_stride", "mod_x", "mod_c")] +
                [(n, _P) for n in ("qkv_w", "qkv_b", "cqkv_w", "cqkv_b", "out_w", "out_b", "cout_w", "cout_b", "qkv2_w", "qkv2_b", "out2_w", "out2_b",
                                   "ff1_w", "ff1_b", "ff2_w", "ff2_b", "cff1_w", "cff1_b", "cff2_w", "cff2_b", "rms_x", "rms_c", "rms_2")])


class MMDiTBlockBwdDesc(ctypes.Structure):
    """advgrpo_mmdit_block_bwd_desc (include/advgrpo.h), field for field."""
    _fields_ = ([(n, c_int32) for n in ("B", "Ni", "Nt", "D", "H", "dual", "last", "first")] + [("mods", _P)] +
                [(n, c_int64) for n in ("mod_stride", "mod_x", "mod_c", "mod_x_prev", "mod_c_prev", "ld_att")] +
                [(n, _P) for n in ("ff2_wT", "ff1_wT", "cff2_wT", "cff1_wT", "out_wT", "cout_wT", "qkv_wT", "cqkv_wT", "out2_wT", "qkv2_wT",
                                   "rms_x", "rms_c", "rms_2",
                                   "x_in", "c_in", "x_mid", "c_mid", "pre", "cpre", "qkv", "rs", "att", "lse", "qkv2", "rs2", "att2", "lse2",
                                   "dx", "dc", "dyg", "dcyg", "dx_out", "dc_out", "dyg_prev", "dcyg_prev", "dyo", "dyc", "dqkv")])


class VaeConv(ctypes.Structure):
    """advgrpo_vae_conv (include/advgrpo.h), field for field."""
    _fields_ = [("w", _P), ("bias", _P), ("cin", c_int32), ("cout", c_int32), ("form", c_int32), ("reserved", c_int32)]


class VaeResnet(ctypes.Structure):
    """advgrpo_vae_resnet (include/advgrpo.h), field for field."""
    _fields_ = [(n, _P) for n in ("norm1_w", "norm1_b", "norm2_w", "norm2_b")] + [("conv1", VaeConv), ("conv2", VaeConv), ("shortcut_w", _P)]


class VaeDecoderDesc(ctypes.Structure):
    """advgrpo_vae_decoder_desc (include/advgrpo.h), field for field."""
    _fields_ = ([(n, c_int32) for n in ("B", "h", "w", "latent_channels", "groups", "n_up", "resnets_per_up", "f16_single")] +
                [("scaling_factor", c_float), ("shift_factor", c_float), ("conv_in", VaeConv), ("conv_out", VaeConv), ("norm_out_w", _P), ("norm_out_b", _P),
                 ("mid", VaeResnet * 2)] +
                [(n, _P) for n in ("attn_norm_w", "attn_norm_b", "attn_q_w", "attn_k_w", "attn_v_w", "attn_o_w", "attn_q_b", "attn_k_b", "attn_v_b", "attn_o_b")] +
                [("up_resnets", POINTER(VaeResnet)), ("upsamplers", POINTER(VaeConv)), ("zero_page", _P)])


class VitLayer(ctypes.Structure):
    """advgrpo_vit_layer (include/advgrpo.h), field for field."""
    _fields_ = [(n, _P) for n in ("ln1_w", "ln1_b", "qkv_w", "qkv_b", "out_w", "out_b", "ls1", "ln2_w", "ln2_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b", "ls2")]


class VitDesc(ctypes.Structure):
    """advgrpo_vit_desc (include/advgrpo.h), field for field."""
    _fields_ = ([(n, c_int32) for n in ("B", "S", "D", "H", "mlp", "n_layers", "act", "causal")] + [("eps", c_float), ("x", _P),
                ("layers", POINTER(VitLayer))])


class LoraMergeItem(ctypes.Structure):
    """advgrpo_lora_merge_item (include/advgrpo.h), field for field."""
    _fields_ = ([(n, _P) for n in ("A", "B", "base", "w", "wT", "a_cat", "b_bd")] + [(n, c_int64) for n in ("ld_base", "ld_w", "ld_wT", "ld_bd")] +
                [("N", c_int32), ("K", c_int32)])


class TnDesc(ctypes.Structure):
    """advgrpo_tn_desc (include/advgrpo.h), field for field."""
    _fields_ = [("P", _P), ("ldp", c_int64), ("p_seg_rows", c_int32), ("p_seg_stride", c_int64), ("p_seg_off", c_int64),
                ("Q", _P), ("ldq", c_int64), ("q_seg_rows", c_int32), ("q_seg_stride", c_int64), ("q_seg_off", c_int64),
                ("C", _P), ("ldc", c_int64), ("transpose_out", c_int32),
                ("M", c_int32), ("N1", c_int32), ("alpha", c_float)]


class Fp8Scales(ctypes.Structure):
    """advgrpo_fp8_scales (include/advgrpo.h)."""
    _fields_ = [("a_scale", _P), ("w_scale", _P)]


# name -> (restype, argtypes); must list every function include/advgrpo.h declares
SIGNATURES = {
    "advgrpo_abi_version": (c_int, []),
    "advgrpo_last_error": (ctypes.c_char_p, []),
    "advgrpo_randn": (c_int, [_P, c_int, c_int64, c_uint64, c_uint64, _P]),
    "advgrpo_sde_step_workspace_bytes": (c_int64, [c_int, c_int64]),
    "advgrpo_sde_step": (c_int, [_P, _P, c_int, c_float, _P, c_int, _P, _P, c_int, c_float, c_int, _P, c_uint64,
                                 c_uint64, _P, c_int, _P, _P, c_int, _P, _P, _P, _P, c_int, c_int64, _P]),
    "advgrpo_sde_step_bwd": (c_int, [_P, _P, c_int, c_float, _P, c_int, _P, _P, c_int, c_float, _P, c_int, _P, _P,
                                     _P, c_int, c_int64, _P]),
    "advgrpo_sde_step_bwd_kl": (c_int, [_P, _P, c_int, c_float, _P, c_int, _P, _P, c_int, c_float, _P, c_int, _P, _P, c_float,
                                        _P, _P, _P, _P, c_int, c_int64, _P]),
    "advgrpo_group_advantage": (c_int, [_P, c_int, _P, c_int, c_int, c_int, _P, _P]),
    "advgrpo_group_advantage_stats": (c_int, [_P, c_int, _P, c_int, c_int, c_int, _P, _P, _P]),
    "advgrpo_grpo_loss": (c_int, [_P, _P, _P, c_int, c_float, c_float, _P, _P, _P]),
    "advgrpo_gemm_bf16": (c_int, [_P, c_int64, _P, c_int64, _P, c_int64, c_int, c_int, c_int, c_int, _P, c_int,
                                  c_float, _P, c_int64, c_int, _P, c_int64, c_int, c_int64, c_int64, c_int, c_int64,
                                  c_int64, c_int, c_int64, c_int64, c_int64, _P]),
    "advgrpo_gemm_variant": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "advgrpo_gemm_grouped": (c_int, [POINTER(GemmDesc), c_int, _P]),
    "advgrpo_gemm_fp8_grouped": (c_int, [POINTER(GemmDesc), POINTER(Fp8Scales), c_int, _P]),
    "advgrpo_quant_fp8_rows": (c_int, [_P, c_int64, _P, c_int64, _P, c_int, c_int, c_int, c_int, _P]),
    "advgrpo_layernorm_mod_pair": (c_int, [POINTER(LnDesc), POINTER(LnDesc), _P]),
    "advgrpo_layernorm_mod": (c_int, [_P, c_int64, _P, _P, c_int64, _P, _P, _P, _P, _P, _P, c_int64, c_int, c_int,
                                      c_int, c_float, _P]),
    "advgrpo_layernorm_mod_fp8": (c_int, [_P, c_int64, _P, _P, c_int64, _P, _P, _P, _P, _P, _P, c_int64, c_int, c_int,
                                          c_int, c_float, _P, _P, _P, _P, c_int64, _P]),
    "advgrpo_rmsnorm_heads": (c_int, [_P, c_int64, c_int, c_int, c_int, _P, c_int, c_float, c_int, c_int64, c_int64,
                                      _P, _P]),
    "advgrpo_rmsnorm_rows": (c_int, [_P, c_int64, _P, c_int64, _P, c_int, c_int, c_float, _P]),
    "advgrpo_rope_half": (c_int, [_P, c_int64, c_int, c_int, c_int, c_int, c_int, _P, _P]),
    "advgrpo_softmax_rows_causal": (c_int, [_P, _P, c_int64, c_int, _P]),
    "advgrpo_qk_norm_rope": (c_int, [_P, c_int64, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, c_int, c_float, _P, _P, _P]),
    "advgrpo_attention_fwd_bias": (c_int, [_P, _P, _P, _P] + [c_int64] * 8 + [c_int] * 5 + [c_float, c_int, _P, _P]),
    "advgrpo_gemm_bf16_train": (c_int, [_P, c_int64, _P, c_int64, _P, c_int64, c_int, c_int, c_int, c_int, _P, c_int,
                                        c_float, _P, c_int64, c_int, _P, c_int64, _P, _P, c_int64, c_int, _P]),
    "advgrpo_gemm_tn_f32acc": (c_int, [_P, c_int64, c_int, c_int64, c_int64, _P, c_int64, c_int, c_int64, c_int64, _P, c_int64,
                                       c_int, c_int, c_int, c_int, c_float, _P, _P]),
    "advgrpo_gemm_tn_workspace_bytes": (c_int64, [c_int, c_int]),
    "advgrpo_mmdit_block_workspace_bytes": (c_int64, [c_int, c_int, c_int, c_int, c_int]),
    "advgrpo_mmdit_block_forward": (c_int, [POINTER(MMDiTBlockDesc), _P, c_int64, _P]),
    "advgrpo_mmdit_block_backward_workspace_bytes": (c_int64, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "advgrpo_mmdit_block_backward": (c_int, [POINTER(MMDiTBlockBwdDesc), _P, c_int64, _P]),
    "advgrpo_vae_decode_workspace_bytes": (c_int64, [POINTER(VaeDecoderDesc)]),
    "advgrpo_vae_decode": (c_int, [POINTER(VaeDecoderDesc), _P, c_int, _P, _P, c_int64, _P]),
    "advgrpo_vit_workspace_bytes": (c_int64, [c_int, c_int, c_int, c_int]),
    "advgrpo_vit_forward": (c_int, [POINTER(VitDesc), _P, c_int64, _P]),
    "advgrpo_gemm_tn_grouped_workspace_bytes": (c_int64, [POINTER(TnDesc), c_int]),
    "advgrpo_gemm_tn_grouped": (c_int, [POINTER(TnDesc), c_int, _P, c_int64, c_int, _P]),
    "advgrpo_lora_merge": (c_int, [_P, c_int, c_int, c_int, c_int, c_float, _P]),
    "advgrpo_transpose_bf16": (c_int, [_P, _P, c_int, c_int, c_int64, c_int64, c_int, c_int, c_int64, c_int64, _P]),
    "advgrpo_layernorm_mod_bwd": (c_int, [_P, c_int64, _P, _P, c_int64, _P, _P, c_int64, c_int, _P, _P, c_int64, c_int,
                                          c_int, c_float, _P]),
    "advgrpo_layernorm_mod_bwd_gated": (c_int, [_P, c_int64, _P, _P, c_int64, _P, _P, c_int64, c_int, _P, _P, c_int64, c_int,
                                                c_int, c_float, _P, _P, _P, _P, c_int64, _P]),
    "advgrpo_rmsnorm_heads_bwd": (c_int, [_P, c_int64, _P, c_int64, _P, c_int, c_int, c_int, _P, c_int, c_int, c_int64,
                                          c_int64, _P]),
    "advgrpo_qk_norm_rope_bwd": (c_int, [_P, c_int64, _P, c_int64, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, c_int, _P, _P]),
    "advgrpo_gate_mul": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int64, _P]),
    "advgrpo_sumsq_workspace_bytes": (c_int64, []),
    "advgrpo_sumsq_f32": (c_int, [_P, c_int64, _P, _P, _P]),
    "advgrpo_adamw_step": (c_int, [_P, _P, _P, _P, _P, c_int64, c_float, c_float, c_float, c_float, c_float, c_int, _P,
                                   c_float, c_float, _P]),
    "advgrpo_ema_step": (c_int, [_P, _P, c_int64, c_float, _P]),
    "advgrpo_gather_rows": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "advgrpo_dino_head_loss": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_float, _P, _P, _P, _P]),
    "advgrpo_cls_attention_fwd": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_float, _P]),
    "advgrpo_cls_attention_bwd": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_float, _P]),
    "advgrpo_clip_pair_loss": (c_int, [_P, _P, c_int, c_int, c_float, _P, _P, _P]),
    "advgrpo_clip_pair_loss_labels": (c_int, [_P, _P, c_int, c_int, c_float, _P, _P, _P, _P, _P]),
    "advgrpo_softmax_bwd_rows": (c_int, [_P, _P, _P, _P, c_int64, c_int, c_int, c_float, _P]),
    "advgrpo_colsum_bf16": (c_int, [_P, c_int64, c_int, c_int, _P, _P]),
    "advgrpo_ln_affine_grads": (c_int, [_P, c_int64, _P, c_int64, c_int, c_int, c_float, _P, _P, _P]),
    "advgrpo_dino_head_dpre": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int, c_int, _P]),
    "advgrpo_timestep_embedding": (c_int, [_P, _P, c_int, c_int, _P]),
    "advgrpo_unary": (c_int, [_P, _P, _P, c_int64, c_int, _P]),
    "advgrpo_patchify": (c_int, [_P, c_int, _P, c_int, c_int, c_int, c_int, _P]),
    "advgrpo_unpatchify": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "advgrpo_conv3x3_nhwc": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P, c_int, _P, _P, _P]),
    "advgrpo_groupnorm_scratch_bytes": (c_int64, [c_int, c_int, c_int]),
    "advgrpo_groupnorm_nhwc": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_float, c_int, _P]),
    "advgrpo_softmax_rows": (c_int, [_P, c_int64, c_int, _P]),
    "advgrpo_latents_to_nhwc": (c_int, [_P, c_int, _P, c_int, c_int, c_int, c_int, c_int, c_float, c_float, _P]),
    "advgrpo_split_bf16x3": (c_int, [_P, _P, _P, c_int64, c_int, c_int, _P]),
    "advgrpo_conv3x3_nhwc_x3": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P, c_int, _P, _P, _P]),
    "advgrpo_latents_mix_to_nhwc": (c_int, [_P, c_int, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P]),
    "advgrpo_rmsnorm_nhwc": (c_int, [_P, c_int, _P, _P, c_int64, c_int, c_float, c_int, c_int, _P]),
    "advgrpo_groupnorm_nhwc_x3": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_float, c_int, c_int, _P]),
    "advgrpo_groupnorm_nhwc_f16x2": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_float, c_int, c_float, _P, c_int, _P]),
    "advgrpo_split_f16x2": (c_int, [_P, _P, _P, c_int64, c_int, c_float, _P]),
    "advgrpo_conv3x3_nhwc_f16x2": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P, c_int, _P, _P, c_float, _P, _P]),
    "advgrpo_conv3x3_nhwc_f16x2_pair": (c_int, [_P, _P, _P, c_float, c_int, c_int, c_int, c_int, c_int, c_int, _P, c_int, _P, _P, c_float, _P]),
    "advgrpo_conv3x3_nhwc_f16x1": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P, c_int, _P, _P, c_float, _P, _P]),
    "advgrpo_conv3x3_nhwc_f16x1_pair": (c_int, [_P, _P, _P, c_float, c_int, c_int, c_int, c_int, c_int, c_int, _P, c_int, _P, _P, c_float, _P]),
    "advgrpo_conv3x3_nhwc_bf16x2": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P, c_int, _P, _P, c_float, _P, _P]),
    "advgrpo_softmax_rows_x3": (c_int, [_P, _P, c_int64, c_int, _P]),
    "advgrpo_layernorm_x3": (c_int, [_P, _P, _P, _P, c_int, c_int, c_float, _P]),
    "advgrpo_split_act_bf16x3": (c_int, [_P, _P, _P, c_int64, c_int, c_int, c_int, _P]),
    "advgrpo_softmax_rows_x3_masked": (c_int, [_P, _P, c_int64, c_int, c_int, c_int, c_float, _P]),
    "advgrpo_add_rows_f32": (c_int, [_P, _P, _P, _P, c_int64, c_int, _P]),
    "advgrpo_latents_to_nhwc_x3": (c_int, [_P, c_int, _P, c_int, c_int, c_int, c_int, c_int, c_float, c_float, _P]),
    "advgrpo_image_postprocess": (c_int, [_P, c_int, c_int, _P, c_int, c_int, c_int, _P]),
    "advgrpo_clip_preprocess_patches": (c_int, [_P, c_int, _P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P, c_int, _P,
                                                _P, c_int, POINTER(c_float), POINTER(c_float), c_int, _P]),
    "advgrpo_clip_preprocess_patches_x3": (c_int, [_P, c_int, _P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P, c_int, _P,
                                                   _P, c_int, POINTER(c_float), POINTER(c_float), c_int, _P]),
    "advgrpo_dino_preprocess_patches": (c_int, [_P, c_int, _P, c_int, c_int, c_int, c_int, c_int, POINTER(c_float),
                                                POINTER(c_float), _P]),
    "advgrpo_dino_preprocess_patches_x3": (c_int, [_P, c_int, _P, c_int, c_int, c_int, c_int, c_int, POINTER(c_float),
                                                   POINTER(c_float), _P]),
    "advgrpo_gather_l2norm_rows": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_float, _P]),
    "advgrpo_dino_head_combine": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_float, _P, _P, _P, _P]),
    "advgrpo_pickscore_pairs": (c_int, [_P, _P, c_int, c_int, c_float, _P, _P]),
    "advgrpo_attention_fwd": (c_int, [_P, _P, _P, _P] + [c_int64] * 8 + [c_int] * 5 + [c_float, c_int, _P, _P]),
    "advgrpo_attention_fallback_count": (c_int, [POINTER(ctypes.c_longlong), c_int]),
    "advgrpo_attention_bwd": (c_int, [_P] * 10 + [c_int64] * 12 + [c_int] * 5 + [c_float, _P]),
}


class AdvGrpoError(RuntimeError):
    pass


_lib = None


def load():
    """Load (once) and return the ctypes handle; raises if the HIP library is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AdvGrpoError(
            f"{LIB_PATH} not found: build the gfx950 kernels first "
            "(python -c 'import __graft_entry__ as g; g.build()' or make -C adv_grpo_amd/csrc). "
            "adv_grpo_amd has no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing: intended
        fn.restype = res
        fn.argtypes = args
    if lib.advgrpo_abi_version() != 1:
        raise AdvGrpoError(f"ABI version mismatch: library {lib.advgrpo_abi_version()} != binding 1")
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise AdvGrpoError(load().advgrpo_last_error().decode() or f"advgrpo call failed ({rc})")


def dtype_code(t: torch.dtype) -> int:
    if t == torch.float32:
        return F32
    if t == torch.bfloat16:
        return BF16
    if t == torch.float64:
        return F64
    raise AdvGrpoError(f"unsupported dtype {t}")


def ptr(t):
    """Device pointer of a contiguous CUDA(HIP) tensor, or None."""
    if t is None:
        return None
    if not t.is_cuda:
        raise AdvGrpoError("adv_grpo_amd kernels take device tensors (got a CPU tensor); there is no CPU path")
    if not t.is_contiguous():
        raise AdvGrpoError("tensor must be contiguous")
    return t.data_ptr()


def stream_ptr(stream=None):
    s = stream if stream is not None else torch.cuda.current_stream()
    return s.cuda_stream
