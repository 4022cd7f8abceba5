"""SD3 / SD3.5 MMDiT ("MMDiT-X") forward in plain torch (test infrastructure).

PARITY UNPINNED: restates diffusers==0.33.1 ``SD3Transformer2DModel`` (the object behind the
reference call sites adv_grpo/diffusers_patch/sd3_pipeline_with_logprob_fast.py:630-637 and
scripts/train_sd3_fast_pickscore.py:235-255) from its published architecture; diffusers is an
un-vendored dependency absent from /root/reference and from this image.
Weights are a flat dict keyed by the diffusers state_dict names, so a real checkpoint can be
dropped in to re-verify.  Runs in the dtype of the weights (fp32 for the oracle).
"""
import math
from dataclasses import dataclass

import torch
import torch.nn.functional as F


@dataclass
class MMDiTConfig:
    num_layers: int = 24
    num_heads: int = 24
    head_dim: int = 64
    in_channels: int = 16
    out_channels: int = 16
    patch_size: int = 2
    joint_attention_dim: int = 4096
    pooled_projection_dim: int = 2048
    pos_embed_max_size: int = 384
    dual_attention_layers: tuple = tuple(range(13))
    qk_norm: bool = True

    @property
    def dim(self):
        return self.num_heads * self.head_dim


def timestep_sinusoid(t, dim=256):
    half = dim // 2
    freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32, device=t.device) / half)
    a = t.float()[:, None] * freqs[None]
    return torch.cat([torch.cos(a), torch.sin(a)], dim=-1)


def _lin(W, name, x):
    return F.linear(x, W[name + ".weight"], W.get(name + ".bias"))


def _ln(x):
    return F.layer_norm(x, (x.shape[-1],), eps=1e-6)


def _rms(x, w, eps=1e-6):
    dt = x.dtype
    v = x.float().pow(2).mean(-1, keepdim=True)
    x = x.float() * torch.rsqrt(v + eps)
    return x.to(dt) * w


def _heads(x, H):
    B, S, D = x.shape
    return x.view(B, S, H, D // H).transpose(1, 2)


def _attn(W, pfx, cfg, x, c=None, pre_only=False):
    H = cfg.num_heads
    q, k, v = (_heads(_lin(W, f"{pfx}.{n}", x), H) for n in ("to_q", "to_k", "to_v"))
    if cfg.qk_norm:
        q, k = _rms(q, W[f"{pfx}.norm_q.weight"]), _rms(k, W[f"{pfx}.norm_k.weight"])
    if c is not None:
        cq, ck, cv = (_heads(_lin(W, f"{pfx}.{n}", c), H) for n in ("add_q_proj", "add_k_proj", "add_v_proj"))
        if cfg.qk_norm:
            cq, ck = _rms(cq, W[f"{pfx}.norm_added_q.weight"]), _rms(ck, W[f"{pfx}.norm_added_k.weight"])
        q, k, v = torch.cat([q, cq], 2), torch.cat([k, ck], 2), torch.cat([v, cv], 2)
    o = F.scaled_dot_product_attention(q, k, v)
    o = o.transpose(1, 2).reshape(x.shape[0], -1, cfg.dim)
    if c is None:
        return _lin(W, f"{pfx}.to_out.0", o), None
    n_img = x.shape[1]
    xo = _lin(W, f"{pfx}.to_out.0", o[:, :n_img])
    co = None if pre_only else _lin(W, f"{pfx}.to_add_out", o[:, n_img:])
    return xo, co


def _ff(W, pfx, x):
    return _lin(W, f"{pfx}.net.2", F.gelu(_lin(W, f"{pfx}.net.0.proj", x), approximate="tanh"))


def block_forward(W, cfg, i, x, c, temb):
    p = f"transformer_blocks.{i}"
    dual = i in cfg.dual_attention_layers
    pre_only = i == cfg.num_layers - 1
    mod = _lin(W, f"{p}.norm1.linear", F.silu(temb))
    if dual:
        sh_a, sc_a, g_a, sh_m, sc_m, g_m, sh_a2, sc_a2, g_a2 = mod.chunk(9, dim=1)
    else:
        sh_a, sc_a, g_a, sh_m, sc_m, g_m = mod.chunk(6, dim=1)
    nx = _ln(x) * (1 + sc_a[:, None]) + sh_a[:, None]
    if dual:  # SD35AdaLayerNormZeroX: second modulation of the same (block-input) LayerNorm
        nx2 = _ln(x) * (1 + sc_a2[:, None]) + sh_a2[:, None]
    cmod = _lin(W, f"{p}.norm1_context.linear", F.silu(temb))
    if pre_only:
        c_sc, c_sh = cmod.chunk(2, dim=1)
        nc = _ln(c) * (1 + c_sc[:, None]) + c_sh[:, None]
    else:
        c_sh_a, c_sc_a, c_g_a, c_sh_m, c_sc_m, c_g_m = cmod.chunk(6, dim=1)
        nc = _ln(c) * (1 + c_sc_a[:, None]) + c_sh_a[:, None]
    a, ca = _attn(W, f"{p}.attn", cfg, nx, nc, pre_only)
    x = x + g_a[:, None] * a
    if dual:
        a2, _ = _attn(W, f"{p}.attn2", cfg, nx2)
        x = x + g_a2[:, None] * a2
    nx = _ln(x) * (1 + sc_m[:, None]) + sh_m[:, None]
    x = x + g_m[:, None] * _ff(W, f"{p}.ff", nx)
    if pre_only:
        return x, None
    c = c + c_g_a[:, None] * ca
    nc = _ln(c) * (1 + c_sc_m[:, None]) + c_sh_m[:, None]
    c = c + c_g_m[:, None] * _ff(W, f"{p}.ff_context", nc)
    return x, c


def embed(W, cfg, hidden_states, timestep, encoder_hidden_states, pooled):
    dt = W["proj_out.weight"].dtype
    B, _, h, w = hidden_states.shape
    ps = cfg.patch_size
    x = F.conv2d(hidden_states.to(dt), W["pos_embed.proj.weight"], W["pos_embed.proj.bias"], stride=ps)
    x = x.flatten(2).transpose(1, 2)
    hh, ww = h // ps, w // ps
    m = cfg.pos_embed_max_size
    top, left = (m - hh) // 2, (m - ww) // 2
    pe = W["pos_embed.pos_embed"].reshape(1, m, m, -1)[:, top:top + hh, left:left + ww].reshape(1, hh * ww, -1)
    x = x + pe.to(dt)
    t_emb = timestep_sinusoid(timestep, 256).to(dt)
    t_emb = _lin(W, "time_text_embed.timestep_embedder.linear_2",
                 F.silu(_lin(W, "time_text_embed.timestep_embedder.linear_1", t_emb)))
    p_emb = _lin(W, "time_text_embed.text_embedder.linear_2",
                 F.silu(_lin(W, "time_text_embed.text_embedder.linear_1", pooled.to(dt))))
    temb = t_emb + p_emb
    c = _lin(W, "context_embedder", encoder_hidden_states.to(dt))
    return x, c, temb


def mmdit_forward(W, cfg, hidden_states, timestep, encoder_hidden_states, pooled_projections, return_intermediates=False):
    B, _, h, w = hidden_states.shape
    x, c, temb = embed(W, cfg, hidden_states, timestep, encoder_hidden_states, pooled_projections)
    inter = {"x0": x, "c0": c, "temb": temb}
    for i in range(cfg.num_layers):
        x, c = block_forward(W, cfg, i, x, c, temb)
        if return_intermediates:
            inter[f"x{i + 1}"] = x
    sc, sh = _lin(W, "norm_out.linear", F.silu(temb)).chunk(2, dim=1)
    x = _ln(x) * (1 + sc[:, None]) + sh[:, None]
    x = _lin(W, "proj_out", x)
    ps, C = cfg.patch_size, cfg.out_channels
    hh, ww = h // ps, w // ps
    x = x.reshape(B, hh, ww, ps, ps, C)
    x = torch.einsum("nhwpqc->nchpwq", x).reshape(B, C, hh * ps, ww * ps)
    return (x, inter) if return_intermediates else x
