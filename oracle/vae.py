"""SD3 VAE decoder (AutoencoderKL.decode) in plain torch (test infrastructure).

PARITY UNPINNED: restates diffusers==0.33.1 ``AutoencoderKL`` decoder for the SD3 VAE config
(latent 16, block_out_channels [128,256,512,512], layers_per_block 2, GroupNorm 32 eps 1e-6, SiLU,
one mid self-attention, no post_quant_conv) -- the object behind the reference call
adv_grpo/diffusers_patch/sd3_pipeline_with_logprob_fast.py:669; diffusers is absent from
/root/reference and from this image.  Weights: diffusers state_dict names.
"""
from dataclasses import dataclass

import torch
import torch.nn.functional as F


@dataclass
class VaeConfig:
    latent_channels: int = 16
    block_out_channels: tuple = (128, 256, 512, 512)
    layers_per_block: int = 2
    norm_num_groups: int = 32
    scaling_factor: float = 1.5305
    shift_factor: float = 0.0609


def _gn(W, name, x, groups, silu):
    x = F.group_norm(x, groups, W[name + ".weight"], W[name + ".bias"], eps=1e-6)
    return F.silu(x) if silu else x


def _conv(W, name, x, pad=1):
    return F.conv2d(x, W[name + ".weight"], W[name + ".bias"], padding=pad)


def _res(W, p, x, G):
    h = _conv(W, f"{p}.conv1", _gn(W, f"{p}.norm1", x, G, True))
    h = _conv(W, f"{p}.conv2", _gn(W, f"{p}.norm2", h, G, True))
    if f"{p}.conv_shortcut.weight" in W:
        x = _conv(W, f"{p}.conv_shortcut", x, pad=0)
    return x + h


def _attn(W, p, x, G):
    B, C, H, Wd = x.shape
    h = _gn(W, f"{p}.group_norm", x, G, False).view(B, C, H * Wd).transpose(1, 2)
    q = F.linear(h, W[f"{p}.to_q.weight"], W[f"{p}.to_q.bias"])
    k = F.linear(h, W[f"{p}.to_k.weight"], W[f"{p}.to_k.bias"])
    v = F.linear(h, W[f"{p}.to_v.weight"], W[f"{p}.to_v.bias"])
    o = F.scaled_dot_product_attention(q[:, None], k[:, None], v[:, None])[:, 0]
    o = F.linear(o, W[f"{p}.to_out.0.weight"], W[f"{p}.to_out.0.bias"])
    return x + o.transpose(1, 2).reshape(B, C, H, Wd)


def vae_decode(W, cfg, z):
    """z: latents ALREADY rescaled by the caller (z/scaling + shift, PF:667).  Returns [B,3,8h,8w]."""
    G = cfg.norm_num_groups
    x = _conv(W, "decoder.conv_in", z)
    x = _res(W, "decoder.mid_block.resnets.0", x, G)
    x = _attn(W, "decoder.mid_block.attentions.0", x, G)
    x = _res(W, "decoder.mid_block.resnets.1", x, G)
    n = len(cfg.block_out_channels)
    for i in range(n):
        for j in range(cfg.layers_per_block + 1):
            x = _res(W, f"decoder.up_blocks.{i}.resnets.{j}", x, G)
        if i < n - 1:
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
            x = _conv(W, f"decoder.up_blocks.{i}.upsamplers.0.conv", x)
    x = _gn(W, "decoder.conv_norm_out", x, G, True)
    return _conv(W, "decoder.conv_out", x)


def postprocess(image):
    """VaeImageProcessor.postprocess(output_type="pt") -- PF:670."""
    return (image / 2 + 0.5).clamp(0, 1)
