"""Qwen-Image VAE decoder (diffusers ``AutoencoderKLQwenImage``, a Wan-2.1-style causal 3-D VAE) for STILL images, on the gfx950
kernels (host orchestration only).  BASELINE config 5's decode: the Qwen-Image twin of ``vae.decode`` + postprocess at
adv_grpo/diffusers_patch/sd3_pipeline_with_logprob_fast.py:667-670 (the reference itself has no Qwen-Image code: README.md:75,
config/grpo.py:324,330); oracle = oracle/qwen_vae.py (PARITY UNPINNED: restated from the published architecture).

A still image is a one-frame clip.  Every CausalConv3d pads two frames in FRONT and none behind, so only the last temporal tap
of a 3x3x3 kernel meets data: the decoder runs as 2-D convolutions with ``weight[:, :, -1]`` (tests/test_oracle_qwen_vae.py checks
that identity on the oracle's literal 3-D form), and the upsamplers' ``time_conv`` (skipped for the first frame of a clip) never
runs.  What differs from the SD3 decoder (vae.py), whose convolution / attention kernels it shares:
  * per-pixel RMS norm over channels (``advgrpo_rmsnorm_nhwc``: one read, one write, no statistics pass) instead of GroupNorm;
  * de-normalisation by per-channel mean / std and the 1x1x1 ``post_quant_conv`` in one small kernel (``advgrpo_latents_mix_to_nhwc``);
  * widths 384 / 192 / 96: the 96-wide full-resolution stage is carried as 128 channels whose upper 32 are EXACTLY zero (zero
    weight rows / columns, zero bias, zero gamma), so the convolution kernels see the 64-channel multiples they are built for
    -- 1.33 x the K of that stage, the price of not having a 96-channel tile;
  * every upsampler halves the width; the mid attention's q / k / v come from one ``to_qkv`` 1x1 convolution.
Modes as in vae.py: "bf16x3" (default; f32 between kernels, split-bf16 products: the fp32 decode to ~1e-5) and "bf16".
"""
import threading

import torch

from . import ops
from .vae import AutoencoderKLDecoder


def _pad64(c):
    return (c + 63) // 64 * 64


class AutoencoderKLQwenImageDecoder(AutoencoderKLDecoder):
    def __init__(self, state_dict, cfg, device="cuda", mode="bf16x3", bf16_weights=True):
        """bf16_weights (bf16x3 mode): run a 3x3 convolution whose weight tensor is EXACT in bf16 (the released checkpoint is bf16) on
        the two-product bf16x2 kernel -- the third product of the split form would multiply by a weight "lo" of zeros (decided per
        tensor at load time; False: always the three split-bf16 products)."""
        self.bf16_weights = bool(bf16_weights)
        if mode not in ("bf16", "bf16x3"):
            raise ValueError(f"AutoencoderKLQwenImageDecoder: mode must be 'bf16' or 'bf16x3', got {mode!r}")
        self.mode, self.cfg = mode, cfg
        self.dtype = torch.float32
        self.device = torch.device(device)
        self.config = type("Cfg", (), {"latents_mean": cfg.latents_mean, "latents_std": cfg.latents_std, "z_dim": cfg.z_dim})()
        self.two_streams, self.n_streams = True, 2
        self._side, self._side_lock = {}, threading.Lock()
        self.w = {}
        self._load(state_dict)

    # ------------------------------------------------------------------ weights
    def _load(self, sd):
        dev, x3 = self.device, self.mode == "bf16x3"
        f32 = lambda t: t.to(device=dev, dtype=torch.float32).contiguous()
        act = (lambda t: f32(t)) if x3 else (lambda t: t.to(device=dev, dtype=torch.bfloat16).contiguous())
        w = self.w

        def conv3(name, key=None):
            """[Co, Ci, (kt,) 3, 3] -> [Co_pad, 9 Ci_pad] in tap-major order (split along Ci in bf16x3 mode)."""
            v = f32(sd[(key or name) + ".weight"])
            if v.dim() == 5:
                v = v[:, :, -1]                         # the one temporal tap a still image meets
            co, ci = v.shape[:2]
            cop, cip = (co if co < 64 else _pad64(co)), _pad64(ci)
            vp = torch.zeros(cop, cip, 3, 3, dtype=torch.float32, device=dev)
            vp[:co, :ci] = v
            vp = vp.permute(0, 2, 3, 1).contiguous()    # [Co, ky, kx, Ci]
            b = torch.zeros(cop, dtype=torch.float32, device=dev)
            b[:co] = f32(sd[(key or name) + ".bias"])
            w[name + ".bias"] = act(b)
            if x3 and self.bf16_weights and cop >= 128 and torch.equal(vp.to(torch.bfloat16).float(), vp):
                w[name + ".weight@bf16"] = vp.reshape(cop, -1).to(torch.bfloat16)
                return
            w[name + ".weight"] = ops.split_x3(vp, order=1).reshape(cop, -1) if x3 else vp.reshape(cop, -1).to(torch.bfloat16)

        def lin(name, v, b):
            """1x1 convolution -> Linear [Co_pad, Ci_pad]."""
            v = f32(v).reshape(v.shape[0], v.shape[1])
            co, ci = v.shape
            vp = torch.zeros(_pad64(co), _pad64(ci), dtype=torch.float32, device=dev)
            vp[:co, :ci] = v
            bp = torch.zeros(_pad64(co), dtype=torch.float32, device=dev)
            bp[:co] = f32(b)
            w[name + ".weight"] = ops.split_x3(vp, order=1) if x3 else vp.to(torch.bfloat16)
            w[name + ".bias"] = act(bp)

        def gamma(name):
            g = f32(sd[name + ".gamma"]).reshape(-1)
            gp = torch.zeros(_pad64(g.numel()), dtype=torch.float32, device=dev)
            gp[:g.numel()] = g
            w[name + ".gamma"] = gp
            w[name + ".mult"] = float(g.numel()) ** 0.5      # QwenImageRMS_norm.scale = dim ** 0.5 of the REAL width

        def res(p):
            gamma(f"{p}.norm1"); conv3(f"{p}.conv1")
            gamma(f"{p}.norm2"); conv3(f"{p}.conv2")
            if f"{p}.conv_shortcut.weight" in sd:
                lin(f"{p}.conv_shortcut", sd[f"{p}.conv_shortcut.weight"], sd[f"{p}.conv_shortcut.bias"])
                if x3:      # the f32 shortcut's bias rides on conv2's (the split GEMM has no bias operand)
                    w[f"{p}.conv2.bias"] = w[f"{p}.conv2.bias"] + w[f"{p}.conv_shortcut.bias"]

        cfg = self.cfg
        C = cfg.z_dim
        w["latents.inv_std"] = f32(1.0 / torch.tensor(cfg.latents_std, dtype=torch.float32))      # the pipeline's `1.0 / latents_std`
        w["latents.mean"] = f32(torch.tensor(cfg.latents_mean, dtype=torch.float32))
        w["post_quant_conv.weight"] = f32(sd["post_quant_conv.weight"]).reshape(C, C).contiguous()
        w["post_quant_conv.bias"] = f32(sd["post_quant_conv.bias"])
        conv3("decoder.conv_in")
        res("decoder.mid_block.resnets.0")
        a = "decoder.mid_block.attentions.0"
        gamma(f"{a}.norm")
        qkv_w, qkv_b = f32(sd[f"{a}.to_qkv.weight"]), f32(sd[f"{a}.to_qkv.bias"])
        D = qkv_w.shape[1]
        for i, n in enumerate(("to_q", "to_k", "to_v")):
            wi = qkv_w[i * D:(i + 1) * D].reshape(D, D).contiguous()
            w[f"{a}.{n}.weight"] = ops.split_x3(wi, order=1) if x3 else wi.to(torch.bfloat16)
            w[f"{a}.{n}.bias"] = act(qkv_b[i * D:(i + 1) * D])
        pw = f32(sd[f"{a}.proj.weight"]).reshape(D, D).contiguous()
        w[f"{a}.to_out.0.weight"] = ops.split_x3(pw, order=1) if x3 else pw.to(torch.bfloat16)
        w[f"{a}.to_out.0.bias"] = act(f32(sd[f"{a}.proj.bias"]))
        res("decoder.mid_block.resnets.1")
        for i in range(len(cfg.dim_mult)):
            for j in range(cfg.num_res_blocks + 1):
                res(f"decoder.up_blocks.{i}.resnets.{j}")
            if cfg.up_block_io(i)[2]:
                conv3(f"decoder.up_blocks.{i}.upsamplers.0.conv", key=f"decoder.up_blocks.{i}.upsamplers.0.resample.1")
        gamma("decoder.norm_out")
        conv3("decoder.conv_out")

    # ------------------------------------------------------------------ pieces
    def _norm(self, name, x, silu, out):
        return ops.rmsnorm_nhwc(x, self.w[name + ".gamma"], self.w[name + ".mult"], silu=silu, out=out)

    def _conv3(self, name, a, **kw):
        """3x3 convolution of split rows `a` ([hi | . | lo] bf16 pieces) in whichever arithmetic the weight allows."""
        w = self.w
        if name + ".weight@bf16" in w:
            return ops.conv3x3_f16x2(a, w[name + ".weight@bf16"], bias=w[name + ".bias"], bf16_pieces=True, **kw)
        return ops.conv3x3_x3(a, w[name + ".weight"], bias=w[name + ".bias"], **kw)

    def _qres3(self, p, x):
        w = self.w
        h = self._conv3(f"{p}.conv1", self._norm(f"{p}.norm1", x, True, "x3pair"))
        sc = x
        if f"{p}.conv_shortcut.weight" in w:
            B, H, W, C = x.shape
            sc = ops.gemm(ops.split_x3(x).view(-1, 3 * C), w[f"{p}.conv_shortcut.weight"], out_dtype=torch.float32).view(B, H, W, -1)
        return self._conv3(f"{p}.conv2", self._norm(f"{p}.norm2", h, True, "x3pair"), residual=sc)

    def _qres(self, p, x):
        w = self.w
        h = ops.conv3x3(self._norm(f"{p}.norm1", x, True, "bf16"), w[f"{p}.conv1.weight"], bias=w[f"{p}.conv1.bias"])
        sc = x
        if f"{p}.conv_shortcut.weight" in w:
            B, H, W, C = x.shape
            sc = ops.gemm(x.view(-1, C), w[f"{p}.conv_shortcut.weight"], bias=w[f"{p}.conv_shortcut.bias"]).view(B, H, W, -1)
        return ops.conv3x3(self._norm(f"{p}.norm2", h, True, "bf16"), w[f"{p}.conv2.weight"], bias=w[f"{p}.conv2.bias"], residual=sc)

    def _head(self, latents, x3):
        w = self.w
        return ops.latents_mix_to_nhwc(latents, 64, w["latents.inv_std"], w["latents.mean"], w["post_quant_conv.weight"],
                                       w["post_quant_conv.bias"], x3=x3)

    def _decode_x3_chain(self, latents):
        cfg, w = self.cfg, self.w
        x = self._conv3("decoder.conv_in", self._head(latents, True))
        x = self._qres3("decoder.mid_block.resnets.0", x)
        a = "decoder.mid_block.attentions.0"
        x = self._attn3_core(a, self._norm(f"{a}.norm", x, False, "x3"), x)
        x = self._qres3("decoder.mid_block.resnets.1", x)
        for i in range(len(cfg.dim_mult)):
            for j in range(cfg.num_res_blocks + 1):
                x = self._qres3(f"decoder.up_blocks.{i}.resnets.{j}", x)
            if cfg.up_block_io(i)[2]:
                u = f"decoder.up_blocks.{i}.upsamplers.0.conv"
                x = self._conv3(u, ops.split_x3(x, order=2), upsample=True)
        y = ops.conv3x3_x3(self._norm("decoder.norm_out", x, True, "x3"), w["decoder.conv_out.weight"], bias=w["decoder.conv_out.bias"])
        return ops.image_postprocess(y)          # clamp(y / 2 + 0.5, 0, 1) == postprocess(clamp(y, -1, 1))

    def _decode_bf16(self, latents):
        cfg, w = self.cfg, self.w
        x = ops.conv3x3(self._head(latents, False), w["decoder.conv_in.weight"], bias=w["decoder.conv_in.bias"])
        x = self._qres("decoder.mid_block.resnets.0", x)
        a = "decoder.mid_block.attentions.0"
        x = self._attn_core(a, self._norm(f"{a}.norm", x, False, "bf16"), x)
        x = self._qres("decoder.mid_block.resnets.1", x)
        for i in range(len(cfg.dim_mult)):
            for j in range(cfg.num_res_blocks + 1):
                x = self._qres(f"decoder.up_blocks.{i}.resnets.{j}", x)
            if cfg.up_block_io(i)[2]:
                u = f"decoder.up_blocks.{i}.upsamplers.0.conv"
                x = ops.conv3x3(x, w[u + ".weight"], bias=w[u + ".bias"], upsample=True)
        y = ops.conv3x3(self._norm("decoder.norm_out", x, True, "bf16"), w["decoder.conv_out.weight"], bias=w["decoder.conv_out.bias"],
                        out_dtype=torch.float32)
        return ops.image_postprocess(y)

    @torch.no_grad()
    def decode_to_image(self, latents):
        """latents [B,16,h,w] as the rollout holds them (normalised) -> image [B,3,8h,8w] f32 in [0,1]: de-normalise,
        post_quant_conv, decode frame 0, clamp, postprocess."""
        if self.mode == "bf16x3":
            return self._decode_x3(latents)
        return self._decode_bf16(latents)


def flops_decode(cfg, h, w):
    """Algorithmic FLOPs of one still-image decode of an h x w latent (2-D taps only: what a one-frame clip executes)."""
    d = cfg.dims
    f, px = 0.0, h * w
    conv = lambda ci, co, n: 2.0 * 9 * ci * co * n
    res = lambda ci, co, n: conv(ci, co, n) + conv(co, co, n) + (2.0 * ci * co * n if ci != co else 0.0)
    f += 2.0 * cfg.z_dim ** 2 * px + conv(cfg.z_dim, d[0], px)
    f += 2 * res(d[0], d[0], px) + (2.0 * 4 * d[0] ** 2 * px + 4.0 * px * px * d[0])
    for i in range(len(cfg.dim_mult)):
        ci, co, up = cfg.up_block_io(i)
        f += res(ci, co, px) + cfg.num_res_blocks * res(co, co, px)
        if up:
            px *= 4
            f += conv(co, co // 2, px)
    return f + conv(d[-1], 3, px)
