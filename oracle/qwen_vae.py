"""Qwen-Image VAE decoder (``AutoencoderKLQwenImage.decode``) in plain torch (test infrastructure).

PARITY UNPINNED: BASELINE config 5 ("Qwen-Image base (MMDiT) 1024x1024, G=8, DINO reward") has no code in the reference
(README.md:75, config/grpo.py:324,330); the pipeline call it would make is the Qwen-Image twin of
adv_grpo/diffusers_patch/sd3_pipeline_with_logprob_fast.py:667-670 (rescale latents, ``vae.decode``, postprocess).  The VAE
lives in diffusers (>= 0.35, ``autoencoder_kl_qwenimage.py``: a Wan-2.1-style causal 3-D convolutional autoencoder), absent
from /root/reference and from this image.  This file restates its published decoder from memory, 5-D weights and all, keyed
by the diffusers state_dict names so a real checkpoint can be dropped in (scripts/verify_against_diffusers.py):

  post_quant_conv  CausalConv3d(16, 16, 1)
  decoder.conv_in  CausalConv3d(16, 384, 3)
  decoder.mid_block: resnets.0, attentions.0 (per-frame single-head attention over H x W, RMS norm, to_qkv / proj 1x1
      Conv2d), resnets.1
  decoder.up_blocks.{0..3}: three residual blocks each (RMS norm -> SiLU -> CausalConv3d 3x3x3, twice; CausalConv3d 1x1x1
      shortcut when the width changes), then upsamplers.0 = nearest-exact x2 in (H, W) + Conv2d(dim, dim / 2, 3) for blocks
      0..2 (blocks 0 and 1 also own a `time_conv` that doubles the frame count: the FIRST frame of a clip skips it, and a
      still image is one frame, so it never runs here and its weights are not part of this dict)
      widths: 384 -> 384 | 192 -> 384 | 192 -> 192 | 96 -> 96   (dims [384, 384, 384, 192, 96], halved by each upsampler)
  decoder.norm_out RMS norm, SiLU, decoder.conv_out CausalConv3d(96, 3, 3); the result is clamped to [-1, 1].

RMS norm = F.normalize(x, dim=channels) * sqrt(C) * gamma (no bias, eps 1e-12 on the norm).  CausalConv3d pads two frames in
FRONT of the clip and none behind (H and W symmetrically), so with one frame only the LAST temporal tap of every 3x3x3
kernel meets data: ``causal_conv3d`` below does the padding literally; ``tests/test_oracle_qwen_vae.py`` checks that this equals
the 2-D convolution with ``weight[:, :, -1]`` which the HIP path runs.
"""
from dataclasses import dataclass

import torch
import torch.nn.functional as F


@dataclass
class QwenVaeConfig:
    base_dim: int = 96
    z_dim: int = 16
    dim_mult: tuple = (1, 2, 4, 4)
    num_res_blocks: int = 2
    # Qwen/Qwen-Image vae/config.json
    latents_mean: tuple = (-0.7571, -0.7089, -0.9113, 0.1075, -0.1745, 0.9653, -0.1517, 1.5508,
                           0.4134, -0.0715, 0.5517, -0.3632, -0.1922, -0.9497, 0.2503, -0.2921)
    latents_std: tuple = (2.8184, 1.4541, 2.3275, 2.6558, 1.2196, 1.7708, 2.6052, 2.0743,
                          3.2687, 2.1526, 2.8652, 1.5579, 1.6382, 1.1253, 2.8251, 1.9160)

    @property
    def dims(self):
        m = self.dim_mult
        return [self.base_dim * u for u in (m[-1],) + tuple(reversed(m))]     # [384, 384, 384, 192, 96]

    def up_block_io(self, i):
        """(input width, output width, has upsampler) of decoder.up_blocks[i]."""
        d = self.dims
        cin = d[i] if i == 0 else d[i] // 2
        return cin, d[i + 1], i != len(self.dim_mult) - 1


def causal_conv3d(W, name, x):
    """x [B, C, T, H, W].  QwenImageCausalConv3d: padding (p, p, p) becomes (W: p, p; H: p, p; T: 2p in front, 0 behind)."""
    w, b = W[name + ".weight"], W.get(name + ".bias")
    pt, ph, pw = (w.shape[2] - 1), (w.shape[3] - 1) // 2, (w.shape[4] - 1) // 2
    x = F.pad(x, (pw, pw, ph, ph, pt, 0))
    return F.conv3d(x, w, b)


def rms_norm(x, gamma, channel_dim=1):
    """QwenImageRMS_norm: F.normalize(x, dim=1) * dim**0.5 * gamma (bias = 0)."""
    C = x.shape[channel_dim]
    return F.normalize(x, dim=channel_dim) * (C ** 0.5) * gamma


def _res(W, p, x):
    h = causal_conv3d(W, f"{p}.conv_shortcut", x) if f"{p}.conv_shortcut.weight" in W else x
    x = causal_conv3d(W, f"{p}.conv1", F.silu(rms_norm(x, W[f"{p}.norm1.gamma"])))
    x = causal_conv3d(W, f"{p}.conv2", F.silu(rms_norm(x, W[f"{p}.norm2.gamma"])))
    return x + h


def _attn(W, p, x):
    B, C, T, H, Wd = x.shape
    ident = x
    y = x.permute(0, 2, 1, 3, 4).reshape(B * T, C, H, Wd)
    y = rms_norm(y, W[f"{p}.norm.gamma"])
    qkv = F.conv2d(y, W[f"{p}.to_qkv.weight"], W[f"{p}.to_qkv.bias"])
    qkv = qkv.reshape(B * T, 1, 3 * C, H * Wd).permute(0, 1, 3, 2).contiguous()
    q, k, v = qkv.chunk(3, dim=-1)
    y = F.scaled_dot_product_attention(q, k, v)
    y = y.squeeze(1).permute(0, 2, 1).reshape(B * T, C, H, Wd)
    y = F.conv2d(y, W[f"{p}.proj.weight"], W[f"{p}.proj.bias"])
    return y.view(B, T, C, H, Wd).permute(0, 2, 1, 3, 4) + ident


def _upsample(W, p, x):
    """QwenImageResample 'upsample2d' / 'upsample3d' on the first (only) frame: per frame nearest-exact x2 + Conv2d."""
    B, C, T, H, Wd = x.shape
    y = x.permute(0, 2, 1, 3, 4).reshape(B * T, C, H, Wd)
    y = F.interpolate(y.float(), scale_factor=(2.0, 2.0), mode="nearest-exact").type_as(y)
    y = F.conv2d(y, W[f"{p}.resample.1.weight"], W[f"{p}.resample.1.bias"], padding=1)
    return y.view(B, T, -1, 2 * H, 2 * Wd).permute(0, 2, 1, 3, 4)


def vae_decode(W, cfg, z):
    """z [B, 16, 1, h, w]: latents ALREADY de-normalised by the caller (z * std + mean, below).  Returns [B, 3, 1, 8h, 8w]
    clamped to [-1, 1] (``AutoencoderKLQwenImage._decode``)."""
    assert z.dim() == 5 and z.shape[2] == 1, "one frame: a still image"
    x = causal_conv3d(W, "post_quant_conv", z)
    x = causal_conv3d(W, "decoder.conv_in", x)
    x = _res(W, "decoder.mid_block.resnets.0", x)
    x = _attn(W, "decoder.mid_block.attentions.0", x)
    x = _res(W, "decoder.mid_block.resnets.1", x)
    for i in range(len(cfg.dim_mult)):
        for j in range(cfg.num_res_blocks + 1):
            x = _res(W, f"decoder.up_blocks.{i}.resnets.{j}", x)
        if cfg.up_block_io(i)[2]:
            x = _upsample(W, f"decoder.up_blocks.{i}.upsamplers.0", x)
    x = F.silu(rms_norm(x, W["decoder.norm_out.gamma"]))
    x = causal_conv3d(W, "decoder.conv_out", x)
    return x.clamp(-1.0, 1.0)


def denormalise(cfg, latents):
    """QwenImagePipeline.__call__ before vae.decode: latents / (1 / std) + mean, per channel."""
    mean = torch.tensor(cfg.latents_mean, dtype=latents.dtype, device=latents.device).view(1, -1, 1, 1, 1)
    inv_std = 1.0 / torch.tensor(cfg.latents_std, dtype=latents.dtype, device=latents.device).view(1, -1, 1, 1, 1)
    return latents / inv_std + mean


def decode_to_image(W, cfg, latents):
    """latents [B, 16, h, w] as the rollout holds them (normalised) -> image [B, 3, 8h, 8w] in [0, 1]
    (de-normalise, decode, take frame 0, VaeImageProcessor.postprocess(output_type="pt"))."""
    x = vae_decode(W, cfg, denormalise(cfg, latents[:, :, None]))[:, :, 0]
    return (x / 2 + 0.5).clamp(0, 1)
