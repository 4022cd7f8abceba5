"""Qwen2.5-VL language model on TEXT-ONLY input (the prompt encoder of Qwen-Image, BASELINE config 5) in plain torch (test infrastructure).

The reference has no Qwen-Image code (README.md:75, config/grpo.py:324,330); this is the Qwen-Image twin of ``encode_prompt``
(scripts/train_dreambooth_lora_sd3.py:98-144 as used at TP:628-651).  What QwenImagePipeline._get_qwen_prompt_embeds does (diffusers >= 0.35,
restated): wrap the prompt in the chat template, run ``text_encoder(input_ids, attention_mask, output_hidden_states=True)``, take
``hidden_states[-1]`` (the output of the final RMSNorm), keep each sample's valid tokens, drop the first 34 (the template's system part)
and right-pad the batch with zeros -> prompt_embeds [B, T - 34, 3584] + mask.

The text encoder itself is PINNED: ``tests/test_oracle_qwen_text.py`` compares ``text_model_forward`` with the installed transformers'
``Qwen2_5_VLTextModel`` on seeded random weights (a text-only prompt gives the three multimodal-rotary position streams the same
indices, so M-RoPE reduces to the standard rotary embedding restated here).  Architecture (Qwen2.5-VL-7B: 28 layers, hidden 3584,
28 query heads / 4 key-value heads of 128, SwiGLU 18944, RMSNorm eps 1e-6, rotary theta 1e6, q/k/v biases, no o_proj bias):
  x = embed_tokens[ids];  per layer:  h = RMSNorm(x);  q, k, v = Linear(h) (+ bias);  rotate_half rotary on q, k;  causal attention
  with grouped key-value heads;  x += o_proj(att);  h = RMSNorm(x);  x += down(silu(gate(h)) * up(h));  finally RMSNorm.
Weights: transformers' state_dict names of the text model (``layers.N.self_attn.q_proj.weight`` ...).
"""
from dataclasses import dataclass

import torch
import torch.nn.functional as F

DROP_IDX = 34      # QwenImagePipeline.prompt_template_encode_start_idx


@dataclass
class QwenTextConfig:
    vocab_size: int = 152064
    hidden_size: int = 3584
    intermediate_size: int = 18944
    num_layers: int = 28
    num_heads: int = 28
    num_kv_heads: int = 4
    rms_eps: float = 1e-6
    rope_theta: float = 1e6

    @property
    def head_dim(self):
        return self.hidden_size // self.num_heads


def rms_norm(x, w, eps):
    """Qwen2RMSNorm: statistics in f32, cast back, times weight."""
    dt = x.dtype
    x = x.float()
    x = x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps)
    return w * x.to(dt)


def rotary_tables(cfg, T, device):
    inv = 1.0 / (cfg.rope_theta ** (torch.arange(0, cfg.head_dim, 2, dtype=torch.float32, device=device) / cfg.head_dim))
    ang = torch.outer(torch.arange(T, dtype=torch.float32, device=device), inv)     # [T, hd/2]
    emb = torch.cat([ang, ang], dim=-1)
    return emb.cos(), emb.sin()                                                      # [T, hd] each


def rotate_half(x):
    h = x.shape[-1] // 2
    return torch.cat([-x[..., h:], x[..., :h]], dim=-1)


def text_model_forward(W, cfg, input_ids, attention_mask=None):
    """-> hidden_states[-1] [B, T, hidden] (after the final norm).  attention_mask [B, T] (1 = token): padding keys are masked."""
    B, T = input_ids.shape
    x = W["embed_tokens.weight"][input_ids]
    cos, sin = rotary_tables(cfg, T, x.device)
    cos, sin = cos.to(x.dtype), sin.to(x.dtype)
    H, KV, hd = cfg.num_heads, cfg.num_kv_heads, cfg.head_dim
    mask = torch.full((T, T), float("-inf"), device=x.device).triu(1)[None, None]     # causal
    if attention_mask is not None:
        mask = mask + (1.0 - attention_mask[:, None, None, :].float()) * torch.finfo(torch.float32).min
    for i in range(cfg.num_layers):
        p = f"layers.{i}"
        h = rms_norm(x, W[f"{p}.input_layernorm.weight"], cfg.rms_eps)
        q = F.linear(h, W[f"{p}.self_attn.q_proj.weight"], W[f"{p}.self_attn.q_proj.bias"]).view(B, T, H, hd).transpose(1, 2)
        k = F.linear(h, W[f"{p}.self_attn.k_proj.weight"], W[f"{p}.self_attn.k_proj.bias"]).view(B, T, KV, hd).transpose(1, 2)
        v = F.linear(h, W[f"{p}.self_attn.v_proj.weight"], W[f"{p}.self_attn.v_proj.bias"]).view(B, T, KV, hd).transpose(1, 2)
        q = q * cos + rotate_half(q) * sin
        k = k * cos + rotate_half(k) * sin
        k, v = k.repeat_interleave(H // KV, dim=1), v.repeat_interleave(H // KV, dim=1)
        att = torch.softmax((q @ k.transpose(-1, -2)).float() * hd ** -0.5 + mask, dim=-1).to(q.dtype) @ v
        x = x + F.linear(att.transpose(1, 2).reshape(B, T, H * hd), W[f"{p}.self_attn.o_proj.weight"])
        h = rms_norm(x, W[f"{p}.post_attention_layernorm.weight"], cfg.rms_eps)
        x = x + F.linear(F.silu(F.linear(h, W[f"{p}.mlp.gate_proj.weight"])) * F.linear(h, W[f"{p}.mlp.up_proj.weight"]),
                         W[f"{p}.mlp.down_proj.weight"])
    return rms_norm(x, W["norm.weight"], cfg.rms_eps)


def qwen_prompt_embeds(W, cfg, input_ids, attention_mask, drop_idx=DROP_IDX):
    """_get_qwen_prompt_embeds after tokenisation: -> (prompt_embeds [B, Lmax, hidden], mask [B, Lmax]) with each sample's valid tokens
    minus the first drop_idx, right-padded with zeros."""
    hs = text_model_forward(W, cfg, input_ids, attention_mask)
    seqs = [hs[b][attention_mask[b].bool()][drop_idx:] for b in range(hs.shape[0])]
    L = max(s.shape[0] for s in seqs)
    emb = torch.stack([torch.cat([s, s.new_zeros(L - s.shape[0], s.shape[1])]) for s in seqs])
    msk = torch.stack([torch.cat([torch.ones(s.shape[0], dtype=torch.long, device=hs.device),
                                  torch.zeros(L - s.shape[0], dtype=torch.long, device=hs.device)]) for s in seqs])
    return emb, msk
