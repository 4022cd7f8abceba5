"""ViT reward towers in plain torch (test infrastructure): CLIP (PickScore_v1 = CLIP ViT-H/14) and
DINOv2 ViT-B/14.

Restates the third-party models behind the reference call sites
  * adv_grpo/pickscore_scorer.py:40-44  (transformers CLIPModel.get_image_features / get_text_features)
  * adv_grpo/rewards.py:397, scripts/train_sd3_fast_dino_patch.py:183-184 (timm vit_base_patch14_dinov2
    forward_features)
PINNED against the installed ``transformers`` CLIPModel and Dinov2Model with seeded random weights
(tests/test_oracle_vit.py); timm itself is absent (Dinov2Model is architecture-equivalent).
Weights: CLIP keyed by transformers state_dict names; DINOv2 keyed by timm names.
"""
from dataclasses import dataclass

import torch
import torch.nn.functional as F


@dataclass
class ClipConfig:  # PickScore_v1 / laion CLIP-ViT-H-14
    v_hidden: int = 1280
    v_layers: int = 32
    v_heads: int = 16
    v_mlp: int = 5120
    image_size: int = 224
    patch: int = 14
    t_hidden: int = 1024
    t_layers: int = 24
    t_heads: int = 16
    t_mlp: int = 4096
    vocab: int = 49408
    max_pos: int = 77
    proj: int = 1024
    eos_token_id: int = 49407
    act: str = "gelu"


@dataclass
class DinoConfig:  # vit_base_patch14_dinov2.lvd142m
    hidden: int = 768
    layers: int = 12
    heads: int = 12
    mlp: int = 3072
    image_size: int = 518
    patch: int = 14


def _act(x, kind):
    return x * torch.sigmoid(1.702 * x) if kind == "quick_gelu" else F.gelu(x)


def _mha(x, qw, qb, kw, kb, vw, vb, ow, ob, H, causal=False):
    B, S, D = x.shape
    hd = D // H
    q = F.linear(x, qw, qb).view(B, S, H, hd).transpose(1, 2)
    k = F.linear(x, kw, kb).view(B, S, H, hd).transpose(1, 2)
    v = F.linear(x, vw, vb).view(B, S, H, hd).transpose(1, 2)
    o = F.scaled_dot_product_attention(q, k, v, is_causal=causal)
    return F.linear(o.transpose(1, 2).reshape(B, S, D), ow, ob)


def _clip_layers(W, pfx, x, n_layers, H, act, causal):
    for i in range(n_layers):
        p = f"{pfx}.encoder.layers.{i}"
        h = F.layer_norm(x, (x.shape[-1],), W[f"{p}.layer_norm1.weight"], W[f"{p}.layer_norm1.bias"], 1e-5)
        a = f"{p}.self_attn"
        x = x + _mha(h, W[f"{a}.q_proj.weight"], W[f"{a}.q_proj.bias"], W[f"{a}.k_proj.weight"], W[f"{a}.k_proj.bias"],
                     W[f"{a}.v_proj.weight"], W[f"{a}.v_proj.bias"], W[f"{a}.out_proj.weight"], W[f"{a}.out_proj.bias"],
                     H, causal)
        h = F.layer_norm(x, (x.shape[-1],), W[f"{p}.layer_norm2.weight"], W[f"{p}.layer_norm2.bias"], 1e-5)
        x = x + F.linear(_act(F.linear(h, W[f"{p}.mlp.fc1.weight"], W[f"{p}.mlp.fc1.bias"]), act),
                         W[f"{p}.mlp.fc2.weight"], W[f"{p}.mlp.fc2.bias"])
    return x


def clip_image_features(W, cfg, pixel_values):
    """CLIPModel.get_image_features: [B,3,224,224] normalised pixels -> [B, proj]."""
    p = "vision_model"
    x = F.conv2d(pixel_values, W[f"{p}.embeddings.patch_embedding.weight"], stride=cfg.patch)
    x = x.flatten(2).transpose(1, 2)
    cls = W[f"{p}.embeddings.class_embedding"].expand(x.shape[0], 1, -1)
    x = torch.cat([cls, x], 1) + W[f"{p}.embeddings.position_embedding.weight"][None]
    x = F.layer_norm(x, (cfg.v_hidden,), W[f"{p}.pre_layrnorm.weight"], W[f"{p}.pre_layrnorm.bias"], 1e-5)
    x = _clip_layers(W, p, x, cfg.v_layers, cfg.v_heads, cfg.act, False)
    pooled = F.layer_norm(x[:, 0], (cfg.v_hidden,), W[f"{p}.post_layernorm.weight"], W[f"{p}.post_layernorm.bias"], 1e-5)
    return F.linear(pooled, W["visual_projection.weight"])


def clip_text_features(W, cfg, input_ids):
    """CLIPModel.get_text_features: [B,77] ids -> [B, proj] (causal mask, pooled at the first EOS)."""
    p = "text_model"
    S = input_ids.shape[1]
    x = W[f"{p}.embeddings.token_embedding.weight"][input_ids] + W[f"{p}.embeddings.position_embedding.weight"][:S][None]
    x = _clip_layers(W, p, x, cfg.t_layers, cfg.t_heads, cfg.act, True)
    x = F.layer_norm(x, (cfg.t_hidden,), W[f"{p}.final_layer_norm.weight"], W[f"{p}.final_layer_norm.bias"], 1e-5)
    if cfg.eos_token_id == 2:      # transformers' legacy rule (modeling_clip.py CLIPTextTransformer.forward): argmax of the ids
        eos = input_ids.int().argmax(dim=-1)
    else:
        eos = (input_ids == cfg.eos_token_id).int().argmax(dim=-1)
    pooled = x[torch.arange(x.shape[0]), eos]
    return F.linear(pooled, W["text_projection.weight"])


def dino_forward_features(W, cfg, x):
    """timm VisionTransformer.forward_features for vit_base_patch14_dinov2: [B,3,518,518] -> [B,1370,768]."""
    x = F.conv2d(x, W["patch_embed.proj.weight"], W["patch_embed.proj.bias"], stride=cfg.patch)
    x = x.flatten(2).transpose(1, 2)
    x = torch.cat([W["cls_token"].expand(x.shape[0], -1, -1), x], 1) + W["pos_embed"]
    D, H = cfg.hidden, cfg.heads
    for i in range(cfg.layers):
        p = f"blocks.{i}"
        h = F.layer_norm(x, (D,), W[f"{p}.norm1.weight"], W[f"{p}.norm1.bias"], 1e-6)
        qw, kw, vw = W[f"{p}.attn.qkv.weight"].chunk(3, 0)
        qb, kb, vb = W[f"{p}.attn.qkv.bias"].chunk(3, 0)
        x = x + W[f"{p}.ls1.gamma"] * _mha(h, qw, qb, kw, kb, vw, vb, W[f"{p}.attn.proj.weight"],
                                           W[f"{p}.attn.proj.bias"], H)
        h = F.layer_norm(x, (D,), W[f"{p}.norm2.weight"], W[f"{p}.norm2.bias"], 1e-6)
        x = x + W[f"{p}.ls2.gamma"] * F.linear(F.gelu(F.linear(h, W[f"{p}.mlp.fc1.weight"], W[f"{p}.mlp.fc1.bias"])),
                                               W[f"{p}.mlp.fc2.weight"], W[f"{p}.mlp.fc2.bias"])
    return F.layer_norm(x, (D,), W["norm.weight"], W["norm.bias"], 1e-6)


def dino_from_hf(sd, cfg):
    """Rename a transformers Dinov2Model state_dict to the timm names used above."""
    W = {"patch_embed.proj.weight": sd["embeddings.patch_embeddings.projection.weight"],
         "patch_embed.proj.bias": sd["embeddings.patch_embeddings.projection.bias"],
         "cls_token": sd["embeddings.cls_token"], "pos_embed": sd["embeddings.position_embeddings"],
         "norm.weight": sd["layernorm.weight"], "norm.bias": sd["layernorm.bias"]}
    for i in range(cfg.layers):
        h, p = f"encoder.layer.{i}", f"blocks.{i}"
        a = f"{h}.attention.attention"
        W[f"{p}.attn.qkv.weight"] = torch.cat([sd[f"{a}.query.weight"], sd[f"{a}.key.weight"], sd[f"{a}.value.weight"]])
        W[f"{p}.attn.qkv.bias"] = torch.cat([sd[f"{a}.query.bias"], sd[f"{a}.key.bias"], sd[f"{a}.value.bias"]])
        W[f"{p}.attn.proj.weight"] = sd[f"{h}.attention.output.dense.weight"]
        W[f"{p}.attn.proj.bias"] = sd[f"{h}.attention.output.dense.bias"]
        for n, m in (("norm1", "norm1"), ("norm2", "norm2")):
            W[f"{p}.{m}.weight"], W[f"{p}.{m}.bias"] = sd[f"{h}.{n}.weight"], sd[f"{h}.{n}.bias"]
        W[f"{p}.ls1.gamma"], W[f"{p}.ls2.gamma"] = sd[f"{h}.layer_scale1.lambda1"], sd[f"{h}.layer_scale2.lambda1"]
        W[f"{p}.mlp.fc1.weight"], W[f"{p}.mlp.fc1.bias"] = sd[f"{h}.mlp.fc1.weight"], sd[f"{h}.mlp.fc1.bias"]
        W[f"{p}.mlp.fc2.weight"], W[f"{p}.mlp.fc2.bias"] = sd[f"{h}.mlp.fc2.weight"], sd[f"{h}.mlp.fc2.bias"]
    return W
