"""SD3 VAE decoder on the gfx950 kernels (host orchestration only).

Stands in for ``pipeline.vae.decode(latents, return_dict=False)[0]`` +
``image_processor.postprocess(image, "pt")`` at
adv_grpo/diffusers_patch/sd3_pipeline_with_logprob_fast.py:667-670.  Activations are NHWC bf16 so
the channel axis (the GEMM K axis of the implicit-GEMM convolutions and the GroupNorm group axis) is
contiguous for 16-byte lane accesses; the nearest x2 upsample is folded into the next convolution's
gather, residual adds and biases into the convolution epilogue.

DEVIATION from the reference: it runs the VAE in fp32 (train_sd3_fast_pickscore.py:481); fp32 matrix
math on MI355X is 1/16 of the bf16 MFMA rate (no TF32 on gfx950), which would make decode as long as the
whole rollout.  Two modes:
  mode="bf16x3"  (default: the reference's arithmetic) split-bf16: every f32 operand is carried as hi + lo bf16 halves
                 and each product runs as xh*wh + xh*wl + xl*wh on the bf16 MFMA (3x the flops, ~2^-16 relative error
                 per product; csrc/conv_x3.hip stages the four pieces once per 64 channels); everything between two
                 matrix products -- bias, residual adds, GroupNorm / softmax inputs -- stays f32.  Image within 3e-5 of
                 the fp32 decode.
  mode="bf16"    the fast opt-in: bf16 operands and activations, f32 accumulation, f32/f64 GroupNorm statistics (2.6x
                 faster, image within 2.3e-2); bench.py prices it next to the default.
Measured tolerances of both in tests/test_gpu_vae.py.
"""
import threading

import torch

from . import ops


class _PairRows:
    """fp16-pair operand rows [B,H,W,3C] (thirds [hi | unwritten | lo] of RAW_PRESCALE * y) written by a producing convolution's epilogue."""
    __slots__ = ("rows",)

    def __init__(self, rows):
        self.rows = rows


class AutoencoderKLDecoder:
    f16_single = False               # (class default: subclasses with their own constructor -- the Qwen-Image decoder -- have no such mode)
    def __init__(self, state_dict, cfg, device="cuda", mode="bf16x3", f16_weights=True, f16_single=False):
        """f16_weights (bf16x3 mode): run a 3x3 convolution whose weight tensor is EXACT in fp16 on the two-product f16x2 kernel
        (decided per tensor at load time; False: always the three split-bf16 products).
        f16_single (opt-in, round 6): those convolutions on ONE fp16 product per f32 product instead of two ("f16x1": the activation's fp16 hi
        half only = a TF32-class operand; the reference runs with allow_tf32 = True, config/base.py:22-23 / TP:537-538).  Not the default:
        bench.py prices it as `vae.value_if_tf32_class`; tests/test_gpu_vae.py holds it to the error of a simulated-TF32 decode."""
        if mode not in ("bf16", "bf16x3"):
            raise ValueError(f"AutoencoderKLDecoder: mode must be 'bf16' or 'bf16x3', got {mode!r}")
        self.mode = mode
        self.f16_weights = bool(f16_weights)
        self.f16_single = bool(f16_single)
        self.cfg = cfg
        self.config = type("Cfg", (), {"scaling_factor": cfg.scaling_factor, "shift_factor": cfg.shift_factor})()
        self.dtype = torch.float32          # what the reference's vae.dtype says (PF:668 casts latents to it)
        self.device = torch.device(device)
        self.G = cfg.norm_num_groups
        self.two_streams = True             # bf16x3 mode: decode a batch as two half batches on two streams
        self.n_streams = 2
        self._side, self._side_lock = {}, threading.Lock()
        self.w = {}
        bf = lambda t: t.to(device=self.device, dtype=torch.bfloat16).contiguous()
        if mode == "bf16x3":
            self._load_x3(state_dict)
            return
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
        return self._attn_core(p, self._gn(f"{p}.group_norm", x, False), x)

    def _attn_core(self, p, h, x):
        """Single-head attention over the H x W pixels of `x` given its normalised copy `h` (bf16 NHWC)."""
        B, H, W, C = x.shape
        T = H * W
        h = h.view(B * T, C)
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

    # ------------------------------------------------------------------ split-bf16 mode
    def _load_x3(self, state_dict):
        f32 = lambda t: t.to(device=self.device, dtype=torch.float32).contiguous()
        w = self.w
        for k, v in state_dict.items():
            if not k.startswith("decoder."):
                continue
            if v.dim() == 4 and v.shape[-1] == 3:       # conv3x3 [Co,Ci,3,3] -> [Co, 9, Ci] -> split along Ci
                v = f32(v)
                co, ci = v.shape[:2]
                # a weight that is exact in fp16 (the released SD3 / SD3.5 VAE is an fp16 checkpoint, upcast at TP:481) goes to the
                # two-product kernel as ONE fp16 piece (include/advgrpo.h "f16x2"); anything else keeps the three bf16 products
                if self.f16_weights and co >= 128 and ci % 64 == 0 and torch.equal(v.half().float(), v):
                    w[k + "@f16"] = v.permute(0, 2, 3, 1).contiguous().half().reshape(co, -1)
                    continue
                if ci % 64:
                    v = torch.cat([v, torch.zeros(co, 64 - ci % 64, 3, 3, dtype=v.dtype, device=v.device)], dim=1)
                w[k] = ops.split_x3(v.permute(0, 2, 3, 1).contiguous(), order=1).reshape(co, -1)
            elif v.dim() == 4:                          # conv1x1 -> linear
                w[k] = ops.split_x3(f32(v).reshape(v.shape[0], v.shape[1]), order=1)
            elif v.dim() == 2:                          # attention projections
                w[k] = ops.split_x3(f32(v), order=1)
            else:
                w[k] = f32(v)
        for k in [k for k in w if k.endswith(".conv_shortcut.bias")]:       # the 1x1 shortcut's bias rides on conv2's
            pre = k[:-len(".conv_shortcut.bias")]
            w[pre + ".conv2.bias"] = w[pre + ".conv2.bias"] + w[k]

    def arithmetic(self):
        """What the 3x3 convolutions of this decoder instance run on, as decided per weight tensor at load time:
        {"f16x2": n, "bf16x3": n, "bf16": n, "total": n, "text": "f16x2 (31/33 convs), bf16x3 (2/33)"} (bench.py prints it)."""
        if self.mode == "bf16":
            n = sum(1 for k, v in self.w.items() if k.endswith(".weight") and ".conv" in k and "shortcut" not in k and v.dim() == 2 and
                    v.shape[1] % 9 == 0 and "to_" not in k)
            return {"f16x2": 0, "f16x1": 0, "bf16x3": 0, "bf16": n, "total": n, "text": f"bf16 ({n}/{n} convs)"}
        f16 = sum(1 for k in self.w if k.endswith(".weight@f16"))
        x3 = sum(1 for k, v in self.w.items() if k.endswith(".weight") and v.dim() == 2 and ("conv_in" in k or "conv_out" in k or ".conv1." in k or
                                                                                            ".conv2." in k or ".upsamplers." in k))
        tot = f16 + x3
        fname = "f16x1" if self.f16_single else "f16x2"
        parts = [f"{name} ({n}/{tot}{' convs' if i == 0 else ''})" for i, (name, n) in enumerate(p for p in ((fname, f16), ("bf16x3", x3)) if p[1])]
        return {"f16x2": 0 if self.f16_single else f16, "f16x1": f16 if self.f16_single else 0, "bf16x3": x3, "bf16": 0, "total": tot, "text": ", ".join(parts)}

    def _conv3(self, name, x3, **kw):
        return ops.conv3x3_x3(x3, self.w[name + ".weight"], bias=self.w[name + ".bias"], **kw)

    fused_gn_stats = True            # A/B switch: GroupNorm statistics from the producing convolution's epilogue
    RAW_PRESCALE = 2.0 ** -4         # un-normalised conv inputs (the upsamplers') as fp16 pairs: |x| up to 1e6 stays in range

    fused_pair_out = True            # A/B switch: a resnet's conv2 in front of an upsampler writes the upsampler's operand rows itself

    def _conv_auto(self, name, x, gn=None, pair_for=None, **kw):
        """3x3 convolution of f32 NHWC `x` (after GroupNorm `gn` + SiLU when given) in whichever arithmetic its weight allows:
        fp16-exact weight -> fp16-pair activations, two products (f16x2); otherwise split-bf16, three products.
        `x` may already be fp16-pair rows (a `_PairRows` from a producer called with pair_for=<this convolution's name>): no split pass.
        pair_for: the ONLY reader of the output is the f16x2 convolution of that name -> return its operand rows, not f32."""
        w = self.w
        if isinstance(x, _PairRows):                     # operand rows made by the producing convolution's epilogue
            assert gn is None and name + ".weight@f16" in w
            return ops.conv3x3_f16x2(x.rows, w[name + ".weight@f16"], bias=w[name + ".bias"], alpha=1.0 / self.RAW_PRESCALE,
                                     gn_stats=self.fused_gn_stats, single=self.f16_single, **kw)
        if pair_for is not None and self.fused_pair_out and name + ".weight@f16" in w and pair_for + ".weight@f16" in w and gn is not None:
            a = ops.groupnorm_nhwc_f16x2(x, w[gn + ".weight"], w[gn + ".bias"], self.G, 1e-6, True,
                                         tile_stats=getattr(x, "gn_tile_stats", None) if self.fused_gn_stats else None)
            return _PairRows(ops.conv3x3_f16x2_pair(a, w[name + ".weight@f16"], self.RAW_PRESCALE, bias=w[name + ".bias"], single=self.f16_single, **kw))
        if name + ".weight@f16" in w:
            # (every f16x2 convolution's output is read by a GroupNorm -- the next resnet's norm1 or this resnet's norm2 -- so its
            #  epilogue leaves that norm's per-tile sums (`gn_tile_stats`) and the norm skips its statistics pass over the activations)
            st = self.fused_gn_stats
            if gn is not None:
                a = ops.groupnorm_nhwc_f16x2(x, w[gn + ".weight"], w[gn + ".bias"], self.G, 1e-6, True,
                                             tile_stats=getattr(x, "gn_tile_stats", None) if st else None)
                return ops.conv3x3_f16x2(a, w[name + ".weight@f16"], bias=w[name + ".bias"], gn_stats=st, single=self.f16_single, **kw)
            a = ops.split_f16x2(x, prescale=self.RAW_PRESCALE)
            return ops.conv3x3_f16x2(a, w[name + ".weight@f16"], bias=w[name + ".bias"], alpha=1.0 / self.RAW_PRESCALE, gn_stats=st,
                                     single=self.f16_single, **kw)
        wide = w[name + ".weight"].shape[0] >= 128      # >= 128 output channels: the kernel that reads the hi and lo thirds only
        a = self._gn3(gn, x, True, pair_only=wide) if gn is not None else ops.split_x3(x, order=2 if wide else 0)
        return self._conv3(name, a, **kw)

    def _gn3(self, name, x, silu, pair_only=False):
        return ops.groupnorm_nhwc_x3(x, self.w[name + ".weight"], self.w[name + ".bias"], self.G, 1e-6, silu, pair_only)

    def _res3(self, p, x, pair_for=None):
        h = self._conv_auto(f"{p}.conv1", x, gn=f"{p}.norm1")
        sc = x
        if f"{p}.conv_shortcut.weight" in self.w:
            B, H, W, C = x.shape
            sc = ops.gemm(ops.split_x3(x).view(-1, 3 * C), self.w[f"{p}.conv_shortcut.weight"], out_dtype=torch.float32
                          ).view(B, H, W, -1)
        return self._conv_auto(f"{p}.conv2", h, gn=f"{p}.norm2", residual=sc, pair_for=pair_for)

    def _attn3(self, p, x):
        return self._attn3_core(p, self._gn3(f"{p}.group_norm", x, False), x)

    def _attn3_core(self, p, h3, x):
        """The same over f32 `x` and its normalised copy as split rows `h3` [B,H,W,3C]."""
        B, H, W, C = x.shape
        T = H * W
        w = self.w
        f32 = torch.float32
        h3 = h3.view(B * T, 3 * C)
        q3 = ops.split_x3(ops.gemm(h3, w[f"{p}.to_q.weight"], out_dtype=f32), 0, bias=w[f"{p}.to_q.bias"]).view(B, T, 3 * C)
        k3 = ops.split_x3(ops.gemm(h3, w[f"{p}.to_k.weight"], out_dtype=f32), 1, bias=w[f"{p}.to_k.bias"]).view(B, T, 3 * C)
        # V^T[b] = Wv . h[b]^T; the weight is the [hi|lo|hi] side, the activations the [hi|hi|lo] side: same three products
        vt = ops.bmm_nt(w[f"{p}.to_v.weight"].unsqueeze(0).expand(B, C, 3 * C), h3.view(B, T, 3 * C), out_dtype=f32)
        vt3 = ops.split_x3(vt, 1)                                                                   # [B, C, 3T]
        s = ops.bmm_nt(q3, k3, alpha=C ** -0.5, out_dtype=f32)                                      # [B, T, T] f32
        p3 = ops.softmax_rows_x3(s)                                                                 # [B, T, 3T]
        del s
        o = ops.bmm_nt(p3, vt3, out_dtype=f32)                                                      # [B, T, C]
        o3 = ops.split_x3(o.view(B * T, C), 0, bias=w[f"{p}.to_v.bias"])    # + bv after P.V: rows of P sum to one
        y = ops.gemm(o3, w[f"{p}.to_out.0.weight"], out_dtype=f32)
        return ops.add_rows_f32(y, x.view(B * T, C), bias=w[f"{p}.to_out.0.bias"]).view(B, H, W, C)

    c_decode = True                  # the chain through advgrpo_vae_decode (one C-ABI call per decode); False: launch by launch

    def _c_desc(self):
        """The decoder's weights as advgrpo_vae_decoder_desc (built once: the tensors it points at live in self.w; B / h / w are set per call on a copy)."""
        d = self.__dict__.get("_cdesc")
        if d is not None:
            return d
        with self._side_lock:            # (decodes start from several rollout threads: one builder, and the host arrays it made stay alive)
            return self._c_desc_build()

    def _c_desc_build(self):
        d = self.__dict__.get("_cdesc")
        if d is not None:
            return d
        import ctypes
        from . import _lib
        w, cfg = self.w, self.cfg

        def conv(name):
            c = _lib.VaeConv()
            if name + ".weight@f16" in w:
                t = w[name + ".weight@f16"]
                c.w, c.cin, c.cout, c.form = t.data_ptr(), t.shape[1] // 9, t.shape[0], 1
            else:
                t = w[name + ".weight"]
                c.w, c.cin, c.cout, c.form = t.data_ptr(), t.shape[1] // 27, t.shape[0], 0
            c.bias = w[name + ".bias"].data_ptr()
            return c

        def resnet(r, p):
            r.norm1_w, r.norm1_b = w[f"{p}.norm1.weight"].data_ptr(), w[f"{p}.norm1.bias"].data_ptr()
            r.norm2_w, r.norm2_b = w[f"{p}.norm2.weight"].data_ptr(), w[f"{p}.norm2.bias"].data_ptr()
            r.conv1, r.conv2 = conv(f"{p}.conv1"), conv(f"{p}.conv2")
            sc = w.get(f"{p}.conv_shortcut.weight")
            r.shortcut_w = sc.data_ptr() if sc is not None else None
        n, per = len(cfg.block_out_channels), cfg.layers_per_block + 1
        d = _lib.VaeDecoderDesc()
        d.latent_channels, d.groups, d.n_up, d.resnets_per_up, d.f16_single = cfg.latent_channels, self.G, n, per, int(self.f16_single)
        d.scaling_factor, d.shift_factor = cfg.scaling_factor, cfg.shift_factor
        d.conv_in, d.conv_out = conv("decoder.conv_in"), conv("decoder.conv_out")
        d.norm_out_w, d.norm_out_b = w["decoder.conv_norm_out.weight"].data_ptr(), w["decoder.conv_norm_out.bias"].data_ptr()
        resnet(d.mid[0], "decoder.mid_block.resnets.0")
        resnet(d.mid[1], "decoder.mid_block.resnets.1")
        a = "decoder.mid_block.attentions.0"
        d.attn_norm_w, d.attn_norm_b = w[f"{a}.group_norm.weight"].data_ptr(), w[f"{a}.group_norm.bias"].data_ptr()
        for k, nm in (("q", "to_q"), ("k", "to_k"), ("v", "to_v"), ("o", "to_out.0")):
            setattr(d, f"attn_{k}_w", w[f"{a}.{nm}.weight"].data_ptr())
            setattr(d, f"attn_{k}_b", w[f"{a}.{nm}.bias"].data_ptr())
        ups = (_lib.VaeResnet * (n * per))()
        for i in range(n):
            for j in range(per):
                resnet(ups[i * per + j], f"decoder.up_blocks.{i}.resnets.{j}")
        samp = (_lib.VaeConv * max(1, n - 1))()
        for i in range(n - 1):
            samp[i] = conv(f"decoder.up_blocks.{i}.upsamplers.0.conv")
        d.up_resnets, d.upsamplers = ctypes.cast(ups, ctypes.POINTER(_lib.VaeResnet)), ctypes.cast(samp, ctypes.POINTER(_lib.VaeConv))
        self._czero = ops.zero_page(self.device)
        d.zero_page = self._czero.data_ptr()
        self._cdesc_arrays = (ups, samp)
        self._cdesc = d
        return d

    def _decode_x3_chain_c(self, latents):
        import ctypes
        from . import _lib
        lib = _lib.load()
        B, C, h, wd = latents.shape
        d = _lib.VaeDecoderDesc.from_buffer_copy(self._c_desc())       # per call: decodes run from several host threads
        d.B, d.h, d.w, d.f16_single = B, h, wd, int(self.f16_single)
        z = latents.contiguous()
        up = 2 ** (len(self.cfg.block_out_channels) - 1)               # one x2 upsampler between consecutive up blocks
        img = torch.empty(B, 3, h * up, wd * up, dtype=torch.float32, device=z.device)
        ws = torch.empty(int(lib.advgrpo_vae_decode_workspace_bytes(ctypes.byref(d))), dtype=torch.uint8, device=z.device)
        _lib.check(lib.advgrpo_vae_decode(ctypes.byref(d), z.data_ptr(), _lib.dtype_code(z.dtype), img.data_ptr(), ws.data_ptr(), ws.numel(),
                                          _lib.stream_ptr()))
        return img

    def _decode_x3_chain(self, latents):
        if self.c_decode and self.fused_gn_stats and self.fused_pair_out:
            return self._decode_x3_chain_c(latents)
        cfg = self.cfg
        x = self._conv3("decoder.conv_in", ops.latents_to_nhwc_x3(latents, 64, cfg.scaling_factor, cfg.shift_factor))
        x = self._res3("decoder.mid_block.resnets.0", x)
        x = self._attn3("decoder.mid_block.attentions.0", x)
        x = self._res3("decoder.mid_block.resnets.1", x)
        n = len(cfg.block_out_channels)
        for i in range(n):
            up = f"decoder.up_blocks.{i}.upsamplers.0.conv" if i < n - 1 else None
            for j in range(cfg.layers_per_block + 1):
                # (the block's last resnet feeds the upsampler's convolution and nothing else: its conv2 writes that operand directly)
                x = self._res3(f"decoder.up_blocks.{i}.resnets.{j}", x, pair_for=up if j == cfg.layers_per_block else None)
            if up is not None:
                x = self._conv_auto(up, x, upsample=True)
        y = self._conv3("decoder.conv_out", self._gn3("decoder.conv_norm_out", x, True))
        return ops.image_postprocess(y)

    def prepare_streams(self, streams, also=()):
        """Choose NOW, by measurement (ops.concurrent_stream), the side stream(s) the two-half-batches decode will use when called on each of
        `streams` -- concurrent with that stream and with the streams in `also` (a scoring stream).  A side stream created lazily inside the
        first decode is whatever hardware queue HIP hands out next: round 6 saw the serial bench leg at 423 instead of 385 ms per step when that
        queue happened to be shared (the decode's halves and the reward future then take turns).  Call at construction time (it synchronises the
        device), as PickScoreScorer.prepare_streams."""
        with self._side_lock:
            for st in streams:
                sides = self._side.setdefault(st.cuda_stream, [])
                while len(sides) < self.n_streams - 1:
                    sides.append(ops.concurrent_stream(self.device, [st] + list(also) + sides))

    def set_side_streams(self, stream, sides):
        """Use THESE streams as the side stream(s) of decodes called on `stream` -- for a caller that already owns streams which are idle whenever
        such a decode runs (the Trainer's rollout streams: decodes on the launch stream only happen while no group is in flight).  Round 6, one
        box: with a freshly measured side stream beside two rollout streams and a scoring stream the in-flight schedule lost 2.5 % (377 vs 369 ms
        per step) although every pair of streams measured as concurrent -- five live streams no longer get a hardware pipe each; without any
        preparation the lazily created side stream cost the serial schedule 10 % (425 vs 387 ms)."""
        with self._side_lock:
            self._side[stream.cuda_stream] = list(sides)[:max(0, self.n_streams - 1)]

    def _decode_x3(self, latents):
        """The decoder is one serial chain in which MFMA-bound convolutions (one workgroup per CU, all of its LDS) alternate with
        HBM-bound GroupNorm / split passes that need no LDS at all.  A batch of two or more images is decoded as two half
        batches on two HIP streams: the hardware runs one half's streaming kernels beside the other half's convolutions
        (the images do not interact and every kernel sums in a fixed order: each image is bit-identical to its single-stream
        decode, tests/test_gpu_vae.py)."""
        B = latents.shape[0]
        n = min(B, self.n_streams) if self.two_streams else 1
        if n < 2:
            return self._decode_x3_chain(latents)
        main = torch.cuda.current_stream(self.device)
        with self._side_lock:                                    # side streams per calling stream (rollout threads have their own)
            sides = self._side.setdefault(main.cuda_stream, [])
            while len(sides) < n - 1:
                sides.append(torch.cuda.Stream(device=self.device))
        parts, outs = latents.tensor_split(n), [None] * n
        for k in range(1, n):
            side = sides[k - 1]
            side.wait_stream(main)                               # the latents are complete on the calling stream
            latents.record_stream(side)
            with torch.cuda.stream(side):
                outs[k] = self._decode_x3_chain(parts[k])
        outs[0] = self._decode_x3_chain(parts[0])
        for k in range(1, n):
            main.wait_stream(sides[k - 1])
            outs[k].record_stream(main)
        return torch.cat(outs)

    @torch.no_grad()
    def decode_to_image(self, latents):
        """latents [B,16,h,w] (pre-scaling, as held by the rollout) -> image [B,3,8h,8w] f32 in [0,1]
        (= PF:667-670: rescale, decode, postprocess)."""
        if self.mode == "bf16x3":
            return self._decode_x3(latents)
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
