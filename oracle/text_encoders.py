"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): CPU restatement of the prompt-encoding path.

encode_prompt (adv_grpo/diffusers_patch/train_dreambooth_lora_sd3.py:98-144) calls three third-party text encoders
that are not in /root/reference (transformers==4.54.0 per setup.py): CLIPTextModelWithProjection (CLIP-L, CLIP-G) and
T5EncoderModel (T5-XXL v1.1).  Restated here in plain torch fp32; PINNED in tests/test_oracle_text_encoders.py against
(a) transformers' own modules with small random configs and (b) the reference's encode_prompt run on those modules.

Weights use the transformers state_dict names."""
import math

import torch
import torch.nn.functional as F

from .vit import _clip_layers


def clip_text_hidden_and_pooled(W, n_layers, heads, act, eos_token_id, input_ids):
    """CLIPTextModelWithProjection(ids, output_hidden_states=True): returns (hidden_states[-2], text_embeds) as used by
    _encode_prompt_with_clip (TD3:60-95): penultimate layer output (no final LayerNorm) and the projected pooled state
    (final LayerNorm, token at the first EOS id, text_projection)."""
    p = "text_model"
    S = input_ids.shape[1]
    x = W[f"{p}.embeddings.token_embedding.weight"][input_ids] + W[f"{p}.embeddings.position_embedding.weight"][:S][None]
    pen = _clip_layers(W, p, x, n_layers - 1, heads, act, True)
    last = pen
    i = n_layers - 1
    lp = f"{p}.encoder.layers.{i}"
    # last layer on top of the penultimate output
    Wl = {k.replace(lp, f"{p}.encoder.layers.0"): v for k, v in W.items() if k.startswith(lp)}
    last = _clip_layers(Wl, p, pen, 1, heads, act, True)
    D = last.shape[-1]
    fin = F.layer_norm(last, (D,), W[f"{p}.final_layer_norm.weight"], W[f"{p}.final_layer_norm.bias"], 1e-5)
    if eos_token_id == 2:      # transformers' legacy rule for the original CLIP vocabularies: argmax of the ids
        eos = input_ids.int().argmax(dim=-1)
    else:
        eos = (input_ids == eos_token_id).int().argmax(dim=-1)
    pooled = fin[torch.arange(fin.shape[0]), eos]
    return pen, F.linear(pooled, W["text_projection.weight"])


def t5_relative_position_bucket(rel, num_buckets=32, max_distance=128):
    """T5Attention._relative_position_bucket, bidirectional (encoder)."""
    num_buckets //= 2
    ret = (rel > 0).long() * num_buckets
    n = rel.abs()
    max_exact = num_buckets // 2
    is_small = n < max_exact
    large = max_exact + (torch.log(n.float() / max_exact) / math.log(max_distance / max_exact) * (num_buckets - max_exact)).long()
    large = torch.min(large, torch.full_like(large, num_buckets - 1))
    return ret + torch.where(is_small, n, large)


def t5_position_bias(rel_emb, S, num_buckets=32, max_distance=128):
    """[H, S, S] additive bias from the layer-0 relative_attention_bias embedding [num_buckets, H]."""
    ctx = torch.arange(S)[:, None]
    mem = torch.arange(S)[None, :]
    bucket = t5_relative_position_bucket(mem - ctx, num_buckets, max_distance)
    return rel_emb[bucket].permute(2, 0, 1).contiguous()


def _t5_norm(x, w, eps=1e-6):
    var = x.float().pow(2).mean(-1, keepdim=True)
    return w * (x * torch.rsqrt(var + eps))


def t5_encoder(W, n_layers, heads, d_kv, input_ids, num_buckets=32, max_distance=128):
    """T5EncoderModel(ids)[0] for the v1.1 ("gated-gelu") family: pre-RMSNorm blocks, un-scaled dot-product attention
    with the shared relative position bias, gated GELU(tanh) feed-forward, final RMSNorm; no attention mask
    (the reference passes none: _encode_prompt_with_t5, TD3:19-56)."""
    x = W["shared.weight"][input_ids]
    B, S, D = x.shape
    bias = t5_position_bias(W["encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"], S, num_buckets,
                            max_distance)
    for i in range(n_layers):
        p = f"encoder.block.{i}.layer"
        h = _t5_norm(x, W[f"{p}.0.layer_norm.weight"])
        a = f"{p}.0.SelfAttention"
        q = F.linear(h, W[f"{a}.q.weight"]).view(B, S, heads, d_kv).transpose(1, 2)
        k = F.linear(h, W[f"{a}.k.weight"]).view(B, S, heads, d_kv).transpose(1, 2)
        v = F.linear(h, W[f"{a}.v.weight"]).view(B, S, heads, d_kv).transpose(1, 2)
        sc = q @ k.transpose(-1, -2) + bias[None]
        o = (torch.softmax(sc.float(), -1).to(v.dtype) @ v).transpose(1, 2).reshape(B, S, heads * d_kv)
        x = x + F.linear(o, W[f"{a}.o.weight"])
        h = _t5_norm(x, W[f"{p}.1.layer_norm.weight"])
        f = f"{p}.1.DenseReluDense"
        g = F.gelu(F.linear(h, W[f"{f}.wi_0.weight"]), approximate="tanh") * F.linear(h, W[f"{f}.wi_1.weight"])
        x = x + F.linear(g, W[f"{f}.wo.weight"])
    return _t5_norm(x, W["encoder.final_layer_norm.weight"])


def encode_prompt(clip_l, clip_g, t5, ids_l, ids_g, ids_t5):
    """encode_prompt (TD3:98-144) on pre-tokenised ids.  clip_* = (W, n_layers, heads, act, eos_id); t5 = (W, n_layers,
    heads, d_kv).  Returns (prompt_embeds [B, 77 + S_t5, d_t5], pooled [B, proj_l + proj_g])."""
    pl, pooled_l = clip_text_hidden_and_pooled(*clip_l, ids_l)
    pg, pooled_g = clip_text_hidden_and_pooled(*clip_g, ids_g)
    clip = torch.cat([pl, pg], dim=-1)
    t5e = t5_encoder(*t5, ids_t5)
    clip = F.pad(clip, (0, t5e.shape[-1] - clip.shape[-1]))
    return torch.cat([clip, t5e], dim=-2), torch.cat([pooled_l, pooled_g], dim=-1)
