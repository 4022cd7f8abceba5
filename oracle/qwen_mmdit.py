"""Qwen-Image MMDiT (``QwenImageTransformer2DModel``) forward in plain torch (test infrastructure).

PARITY UNPINNED: BASELINE config 5 ("Qwen-Image base (MMDiT) 1024x1024, G=8, DINO reward, fp8 MFMA path") has no code in
the reference beyond a to-do line (README.md:75 "Try more base models like QWen-Image") and the `pretrained.model` /
`resolution` switches of config/grpo.py:324,330; the model itself lives in diffusers (>= 0.35, newer than the 0.33.1 the
reference pins, setup.py:8-52), which is absent from /root/reference and from this image.  This file restates the published
architecture of diffusers' ``QwenImageTransformer2DModel`` / ``QwenImageTransformerBlock`` /
``QwenDoubleStreamAttnProcessor2_0`` / ``QwenEmbedRope`` (Qwen/Qwen-Image ``transformer/config.json``: 60 layers,
24 heads x 128, in_channels 64 = 2x2-packed 16-channel latents, joint_attention_dim 3584, axes_dims_rope (16, 56, 56)) from
memory.  Weights are a flat dict keyed by the diffusers state_dict names, so a real checkpoint can be dropped in to re-verify
(scripts/verify_against_diffusers.py has the command).  Runs in the dtype of the weights (fp32 for the oracle).

What is restated (names as in diffusers):
  img_in Linear(64 -> D), txt_norm RMSNorm(3584, eps 1e-6) + txt_in Linear(3584 -> D),
  time_text_embed = TimestepEmbedding(Timesteps(256, flip_sin_to_cos, shift 0, scale 1000)(sigma)) -- no pooled projection,
  60 x block: img_mod / txt_mod = Linear(SiLU(temb)) -> 6 D each, chunks (shift, scale, gate) x (attention, MLP);
      LayerNorm(no affine, 1e-6) -> modulate -> to_q/k/v | add_q/k/v_proj -> per-head RMSNorm(128, 1e-6, affine) ->
      rotary embedding on adjacent (even, odd) pairs (complex multiplication by e^{i pos freq}) -> joint attention over
      [text ; image] tokens (no mask, scale 128^-1/2) -> to_out.0 / to_add_out -> gated residual;
      LayerNorm -> modulate -> Linear(D, 4D) GELU(tanh) Linear(4D, D) -> gated residual; both streams, every block,
  norm_out AdaLayerNormContinuous (scale, shift) + proj_out Linear(D -> 64).
"""
import math
from dataclasses import dataclass

import torch
import torch.nn.functional as F


@dataclass
class QwenMMDiTConfig:
    num_layers: int = 60
    num_heads: int = 24
    head_dim: int = 128
    in_channels: int = 64          # 2x2-packed latents of the 16-channel VAE
    out_channels: int = 16
    patch_size: int = 2
    joint_attention_dim: int = 3584
    axes_dims_rope: tuple = (16, 56, 56)
    rope_theta: float = 10000.0
    scale_rope: bool = True

    @property
    def dim(self):
        return self.num_heads * self.head_dim


def _lin(W, name, x):
    return F.linear(x, W[name + ".weight"], W.get(name + ".bias"))


def _ln(x):
    return F.layer_norm(x, (x.shape[-1],), eps=1e-6)


def _rms(x, w, eps=1e-6):
    """diffusers RMSNorm: statistics in f32, cast to the weight dtype, then the affine weight."""
    dt = x.dtype
    v = x.float().pow(2).mean(-1, keepdim=True)
    x = x.float() * torch.rsqrt(v + eps)
    return x.to(dt) * w


def timestep_sinusoid(sigma, dim=256, scale=1000.0):
    """get_timestep_embedding(sigma, 256, flip_sin_to_cos=True, downscale_freq_shift=0, scale=1000): [cos | sin]."""
    half = dim // 2
    freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32, device=sigma.device) / half)
    a = scale * (sigma.float()[:, None] * freqs[None])
    return torch.cat([torch.cos(a), torch.sin(a)], dim=-1)


# ---------------------------------------------------------------- rotary tables (QwenEmbedRope)
def _rope_params(index, dim, theta):
    freqs = torch.outer(index.float(), 1.0 / torch.pow(torch.tensor(theta, dtype=torch.float32),
                                                       torch.arange(0, dim, 2).float().div(dim)))
    return torch.polar(torch.ones_like(freqs), freqs)          # complex64 [len, dim/2]


def rope_freqs(cfg, frame, height, width, txt_len, table=4096):
    """(vid_freqs [frame*height*width, head_dim/2], txt_freqs [txt_len, head_dim/2]) complex64, for ONE image of
    frame x height x width PACKED latent positions (height = latent_h / 2).  scale_rope: the spatial axes are centred
    (negative indices for the first half), and the text positions start after the largest spatial half-extent."""
    ax = cfg.axes_dims_rope
    pos_index = torch.arange(table)
    neg_index = torch.arange(table).flip(0) * -1 - 1
    pos = [_rope_params(pos_index, d, cfg.rope_theta) for d in ax]
    neg = [_rope_params(neg_index, d, cfg.rope_theta) for d in ax]
    f_frame = pos[0][:frame].view(frame, 1, 1, -1).expand(frame, height, width, -1)
    if cfg.scale_rope:
        f_h = torch.cat([neg[1][-(height - height // 2):], pos[1][:height // 2]], dim=0)
        f_w = torch.cat([neg[2][-(width - width // 2):], pos[2][:width // 2]], dim=0)
        max_vid_index = max(height // 2, width // 2)
    else:
        f_h, f_w = pos[1][:height], pos[2][:width]
        max_vid_index = max(height, width)
    f_h = f_h.view(1, height, 1, -1).expand(frame, height, width, -1)
    f_w = f_w.view(1, 1, width, -1).expand(frame, height, width, -1)
    vid = torch.cat([f_frame, f_h, f_w], dim=-1).reshape(frame * height * width, -1)
    txt = torch.cat(pos, dim=1)[max_vid_index:max_vid_index + txt_len]
    return vid.contiguous(), txt.contiguous()


def apply_rope(x, freqs):
    """apply_rotary_emb_qwen(use_real=False): x [B, S, H, hd]; adjacent pairs (2i, 2i+1) are one complex number."""
    xc = torch.view_as_complex(x.float().reshape(*x.shape[:-1], -1, 2))
    out = torch.view_as_real(xc * freqs.to(x.device).unsqueeze(1)).flatten(3)
    return out.type_as(x)


# ---------------------------------------------------------------- block
def _modulate(x, mod):
    shift, scale, gate = mod.chunk(3, dim=-1)
    return x * (1 + scale.unsqueeze(1)) + shift.unsqueeze(1), gate.unsqueeze(1)


def _ff(W, pfx, x):
    return _lin(W, f"{pfx}.net.2", F.gelu(_lin(W, f"{pfx}.net.0.proj", x), approximate="tanh"))


def block_forward(W, cfg, i, x, c, temb, vid_freqs, txt_freqs):
    p = f"transformer_blocks.{i}"
    H = cfg.num_heads
    img_mod1, img_mod2 = _lin(W, f"{p}.img_mod.1", F.silu(temb)).chunk(2, dim=-1)
    txt_mod1, txt_mod2 = _lin(W, f"{p}.txt_mod.1", F.silu(temb)).chunk(2, dim=-1)
    nx, g_x1 = _modulate(_ln(x), img_mod1)
    nc, g_c1 = _modulate(_ln(c), txt_mod1)
    B, Ni, _ = x.shape
    Nt = c.shape[1]
    heads = lambda t: t.unflatten(-1, (H, -1))                       # [B, S, H, hd]
    q, k, v = (heads(_lin(W, f"{p}.attn.{n}", nx)) for n in ("to_q", "to_k", "to_v"))
    cq, ck, cv = (heads(_lin(W, f"{p}.attn.{n}", nc)) for n in ("add_q_proj", "add_k_proj", "add_v_proj"))
    q, k = _rms(q, W[f"{p}.attn.norm_q.weight"]), _rms(k, W[f"{p}.attn.norm_k.weight"])
    cq, ck = _rms(cq, W[f"{p}.attn.norm_added_q.weight"]), _rms(ck, W[f"{p}.attn.norm_added_k.weight"])
    q, k = apply_rope(q, vid_freqs), apply_rope(k, vid_freqs)
    cq, ck = apply_rope(cq, txt_freqs), apply_rope(ck, txt_freqs)
    jq, jk, jv = torch.cat([cq, q], 1), torch.cat([ck, k], 1), torch.cat([cv, v], 1)     # [text ; image]
    o = F.scaled_dot_product_attention(jq.transpose(1, 2), jk.transpose(1, 2), jv.transpose(1, 2))
    o = o.transpose(1, 2).flatten(2, 3).to(q.dtype)
    co, xo = o[:, :Nt], o[:, Nt:]
    x = x + g_x1 * _lin(W, f"{p}.attn.to_out.0", xo)
    c = c + g_c1 * _lin(W, f"{p}.attn.to_add_out", co)
    nx, g_x2 = _modulate(_ln(x), img_mod2)
    x = x + g_x2 * _ff(W, f"{p}.img_mlp", nx)
    nc, g_c2 = _modulate(_ln(c), txt_mod2)
    c = c + g_c2 * _ff(W, f"{p}.txt_mlp", nc)
    return x, c


def pack_latents(lat):
    """QwenImagePipeline._pack_latents: [B, C, h, w] -> [B, (h/2)(w/2), 4C], column = c*4 + py*2 + px."""
    B, C, h, w = lat.shape
    return lat.view(B, C, h // 2, 2, w // 2, 2).permute(0, 2, 4, 1, 3, 5).reshape(B, (h // 2) * (w // 2), C * 4)


def unpack_latents(tok, h, w):
    """QwenImagePipeline._unpack_latents (one frame): [B, (h/2)(w/2), 4C] -> [B, C, h, w]."""
    B, _, C4 = tok.shape
    return tok.view(B, h // 2, w // 2, C4 // 4, 2, 2).permute(0, 3, 1, 4, 2, 5).reshape(B, C4 // 4, h, w)


def qwen_forward(W, cfg, hidden_states, sigma, encoder_hidden_states, return_intermediates=False):
    """hidden_states [B, 16, h, w] latents (packed here as the pipeline does), sigma [B] in [0, 1] (= timestep / 1000, what
    QwenImagePipeline passes), encoder_hidden_states [B, Nt, 3584] -> velocity [B, 16, h, w]."""
    dt = W["proj_out.weight"].dtype
    B, _, h, w = hidden_states.shape
    tok = pack_latents(hidden_states.to(dt))
    x = _lin(W, "img_in", tok)
    c = _rms(encoder_hidden_states.to(dt), W["txt_norm.weight"])
    c = _lin(W, "txt_in", c)
    temb = _lin(W, "time_text_embed.timestep_embedder.linear_2",
                F.silu(_lin(W, "time_text_embed.timestep_embedder.linear_1", timestep_sinusoid(sigma).to(dt))))
    vid_freqs, txt_freqs = rope_freqs(cfg, 1, h // 2, w // 2, encoder_hidden_states.shape[1])
    inter = {"x0": x, "c0": c, "temb": temb}
    for i in range(cfg.num_layers):
        x, c = block_forward(W, cfg, i, x, c, temb, vid_freqs, txt_freqs)
        if return_intermediates:
            inter[f"x{i + 1}"] = x
            inter[f"c{i + 1}"] = c
    sc, sh = _lin(W, "norm_out.linear", F.silu(temb)).chunk(2, dim=1)
    x = _ln(x) * (1 + sc[:, None]) + sh[:, None]
    out = unpack_latents(_lin(W, "proj_out", x), h, w)
    return (out, inter) if return_intermediates else out
