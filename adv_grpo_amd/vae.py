"""SD3 VAE decoder on the gfx950 kernels (host orchestration only).

Stands in for ``pipeline.vae.decode(latents, return_dict=False)[0]`` +
``image_processor.postprocess(image, "pt")`` at
adv_grpo/diffusers_patch/sd3_pipeline_with_logprob_fast.py:667-670.  Activations are NHWC bf16 so
the channel axis (the GEMM K axis of the implicit-GEMM convolutions and the GroupNorm group axis) is
contiguous for 16-byte lane accesses; the nearest x2 upsample is folded into the next convolution's
gather, residual adds and biases into the convolution epilogue.

DEVIATION from the reference: it runs the VAE in fp32 (train_sd3_fast_pickscore.py:481); fp32 matrix
math on MI355X is 1/16 of the bf16 MFMA rate (no TF32 on gfx950), which would make decode as long as the
whole rollout.  Here: bf16 operands, f32 accumulation, f32/f64 GroupNorm statistics.  Measured
tolerance in tests/test_gpu_vae.py.
"""
import torch

from . import ops


class AutoencoderKLDecoder:
    def __init__(self, state_dict, cfg, device="cuda"):
        self.cfg = cfg
        self.config = type("Cfg", (), {"scaling_factor": cfg.scaling_factor, "shift_factor": cfg.shift_factor})()
        self.dtype = torch.float32          # what the reference's vae.dtype says (PF:668 casts latents to it)
        self.device = torch.device(device)
        self.G = cfg.norm_num_groups
        self.w = {}
        bf = lambda t: t.to(device=self.device, dtype=torch.bfloat16).contiguous()
        for k, v in state_dict.items():
            if not k.startswith("decoder."):
                continue
            if v.dim() == 4 and v.shape[-1] == 3:       # conv3x3 [Co,Ci,3,3] -> [Co, (ky,kx,ci)]
                co, ci = v.shape[:2]
                if ci % 64:                              # conv_in: pad 16 -> 64 input channels
                    pad = torch.zeros(co, 64 - ci % 64, 3, 3, dtype=v.dtype, device=v.device)
                    v = torch.cat([v, pad], dim=1)
                self.w[k] = bf(v.permute(0, 2, 3, 1).reshape(co, -1))
            elif v.dim() == 4:                          # conv1x1 -> linear
                self.w[k] = bf(v.reshape(v.shape[0], v.shape[1]))
            else:
                self.w[k] = bf(v)

    def _conv(self, name, x, **kw):
        return ops.conv3x3(x, self.w[name + ".weight"], bias=self.w[name + ".bias"], **kw)

    def _gn(self, name, x, silu):
        return ops.groupnorm_nhwc(x, self.w[name + ".weight"], self.w[name + ".bias"], self.G, 1e-6, silu)

    def _res(self, p, x):
        h = self._conv(f"{p}.conv1", self._gn(f"{p}.norm1", x, True))
        sc = x
        if f"{p}.conv_shortcut.weight" in self.w:
            B, H, W, C = x.shape
            sc = ops.gemm(x.view(-1, C), self.w[f"{p}.conv_shortcut.weight"], bias=self.w[f"{p}.conv_shortcut.bias"]
                          ).view(B, H, W, -1)
        return self._conv(f"{p}.conv2", self._gn(f"{p}.norm2", h, True), residual=sc)

    def _attn(self, p, x):
        B, H, W, C = x.shape
        T = H * W
        h = self._gn(f"{p}.group_norm", x, False).view(B * T, C)
        w = self.w
        q = ops.gemm(h, w[f"{p}.to_q.weight"], bias=w[f"{p}.to_q.bias"]).view(B, T, C)
        k = ops.gemm(h, w[f"{p}.to_k.weight"], bias=w[f"{p}.to_k.bias"]).view(B, T, C)
        # V^T[b] = Wv . h[b]^T (the bias is added after P.V: rows of P sum to one)
        vt = ops.bmm_nt(w[f"{p}.to_v.weight"].unsqueeze(0).expand(B, C, C), h.view(B, T, C))      # [B, C, T]
        s = ops.bmm_nt(q, k, alpha=C ** -0.5)                                                      # [B, T, T]
        ops.softmax_rows_(s)
        o = ops.bmm_nt(s, vt)                                                                      # [B, T, C]
        o = ops.unary(o.view(-1), None, x2=w[f"{p}.to_v.bias"].repeat(B * T)).view(B * T, C)
        y = ops.gemm(o, w[f"{p}.to_out.0.weight"], bias=w[f"{p}.to_out.0.bias"], residual=x.view(B * T, C))
        return y.view(B, H, W, C)

    @torch.no_grad()
    def decode_to_image(self, latents):
        """latents [B,16,h,w] (pre-scaling, as held by the rollout) -> image [B,3,8h,8w] f32 in [0,1]
        (= PF:667-670: rescale, decode, postprocess)."""
        cfg = self.cfg
        x = ops.latents_to_nhwc(latents, 64, cfg.scaling_factor, cfg.shift_factor)
        x = self._conv("decoder.conv_in", x)
        x = self._res("decoder.mid_block.resnets.0", x)
        x = self._attn("decoder.mid_block.attentions.0", x)
        x = self._res("decoder.mid_block.resnets.1", x)
        n = len(cfg.block_out_channels)
        for i in range(n):
            for j in range(cfg.layers_per_block + 1):
                x = self._res(f"decoder.up_blocks.{i}.resnets.{j}", x)
            if i < n - 1:
                x = self._conv(f"decoder.up_blocks.{i}.upsamplers.0.conv", x, upsample=True)
        x = self._gn("decoder.conv_norm_out", x, True)
        y = self._conv("decoder.conv_out", x, out_dtype=torch.float32)
        return ops.image_postprocess(y)
