"""CLIP ViT-H/14 (PickScore_v1) towers in fp32-equivalent arithmetic on the bf16 matrix units: the fp32 scorer the reference
builds for its `pickscore` reward (adv_grpo/rewards.py:561-574 -> PickScoreScorer(dtype=torch.float32), config 4's reward and
the eval reward of the PickScore configs).

Every matrix product runs as three bf16 MFMA products with f32 accumulation on split operands (hi = bf16(v), lo = bf16(v - hi);
x w = xh wh + xh wl + xl wh, the dropped xl wl term is 2^-16 relative: include/advgrpo.h "bf16x3"), everything between two
products -- residual stream, LayerNorm statistics, softmax, GELU, biases -- is f32 (csrc/x3.hip row kernels).  Same weights,
same state-dict names and same call surface as vit.CLIPModel; features come back in f32.  The towers are small next to the
rollout (0.38 TFLOP per image, three times that here): nothing is tuned, the per-head score matrices are materialised.
"""
import torch

from . import ops, preprocess

f32 = torch.float32


def _pad64(n):
    return (n + 63) // 64 * 64


def _w3(w, dev):
    """Linear weight [N, K] f32 -> right operand [N, 3K] bf16 ([hi | lo | hi])."""
    return ops.split_x3(w.to(device=dev, dtype=f32).contiguous(), 1)


def _v(t, dev):
    return t.to(device=dev, dtype=f32).contiguous()


def linear_x3(x3, w3):
    """x3 [M, 3K] split rows, w3 [N, 3K] -> f32 [M, N] (bias added by the consumer)."""
    return ops.gemm(x3, w3, out_dtype=f32)


class _EncoderX3:
    """Pre-LN transformer encoder stack, f32 residual stream [B*S, D]."""

    def __init__(self, layers, heads, eps, act, causal=False):
        self.layers, self.heads, self.eps, self.act, self.causal = layers, heads, eps, act, causal

    def _heads(self, t, B, S, Sp, dp):
        """[B, S, H, d] f32 -> zero-padded [B*H, Sp, dp]."""
        H, d = t.shape[2], t.shape[3]
        out = torch.zeros(B * H, Sp, dp, dtype=f32, device=t.device)
        out.view(B, H, Sp, dp)[:, :, :S, :d] = t.permute(0, 2, 1, 3)
        return out

    def __call__(self, x, B, S):
        D = x.shape[1]
        H = self.heads
        d = D // H
        Sp, dp = _pad64(S), _pad64(d)                 # contraction lengths 3*dp / 3*Sp must be multiples of 64
        for L in self.layers:
            h3 = ops.layernorm_x3(x, L["ln1.w"], L["ln1.b"], self.eps)
            qkv = ops.add_rows_f32(linear_x3(h3, L["qkv.w3"]), None, L["qkv.b"]).view(B, S, 3, H, d)
            q, k, v = (self._heads(qkv[:, :, j], B, S, Sp, dp) for j in range(3))
            q3 = ops.split_x3(q.view(-1, dp), 0).view(B * H, Sp, 3 * dp)
            k3 = ops.split_x3(k.view(-1, dp), 1).view(B * H, Sp, 3 * dp)
            s = ops.bmm_nt(q3, k3, out_dtype=f32)                                           # [BH, Sp, Sp]
            p3 = ops.softmax_rows_x3_masked(s.view(-1, Sp), S, causal_period=Sp if self.causal else 0, alpha=d ** -0.5)
            vt3 = ops.split_x3(v.transpose(1, 2).contiguous().view(-1, Sp), 1).view(B * H, dp, 3 * Sp)
            o = ops.bmm_nt(p3.view(B * H, Sp, 3 * Sp), vt3, out_dtype=f32)                  # [BH, Sp, dp]
            o = o.view(B, H, Sp, dp)[:, :, :S, :d].permute(0, 2, 1, 3).reshape(B * S, D).contiguous()
            x = self._residual(x, linear_x3(ops.split_x3(o, 0), L["out.w3"]), L["out.b"], L.get("ls1"))
            h3 = ops.layernorm_x3(x, L["ln2.w"], L["ln2.b"], self.eps)
            m3 = ops.split_act_x3(linear_x3(h3, L["fc1.w3"]), 0, bias=L["fc1.b"], act=self.act)
            x = self._residual(x, linear_x3(m3, L["fc2.w3"]), L["fc2.b"], L.get("ls2"))
        return x

    @staticmethod
    def _residual(x, y, bias, layer_scale):
        """x + (y + bias) [* gamma: DINOv2's LayerScale], f32."""
        if layer_scale is None:
            return ops.add_rows_f32(x, y, bias)
        return torch.addcmul(x, ops.add_rows_f32(y, None, bias), layer_scale)


def pack_clip_layers_x3(sd, pfx, n, dev):
    out = []
    for i in range(n):
        p = f"{pfx}.encoder.layers.{i}"
        a = f"{p}.self_attn"
        out.append({
            "ln1.w": _v(sd[f"{p}.layer_norm1.weight"], dev), "ln1.b": _v(sd[f"{p}.layer_norm1.bias"], dev),
            "ln2.w": _v(sd[f"{p}.layer_norm2.weight"], dev), "ln2.b": _v(sd[f"{p}.layer_norm2.bias"], dev),
            "qkv.w3": _w3(torch.cat([sd[f"{a}.q_proj.weight"], sd[f"{a}.k_proj.weight"], sd[f"{a}.v_proj.weight"]]), dev),
            "qkv.b": _v(torch.cat([sd[f"{a}.q_proj.bias"], sd[f"{a}.k_proj.bias"], sd[f"{a}.v_proj.bias"]]), dev),
            "out.w3": _w3(sd[f"{a}.out_proj.weight"], dev), "out.b": _v(sd[f"{a}.out_proj.bias"], dev),
            "fc1.w3": _w3(sd[f"{p}.mlp.fc1.weight"], dev), "fc1.b": _v(sd[f"{p}.mlp.fc1.bias"], dev),
            "fc2.w3": _w3(sd[f"{p}.mlp.fc2.weight"], dev), "fc2.b": _v(sd[f"{p}.mlp.fc2.bias"], dev)})
    return out


class CLIPModelX3:
    """vit.CLIPModel's surface (get_image_features / get_text_features / logit_scale), fp32-equivalent arithmetic."""

    def __init__(self, sd, cfg, device="cuda"):
        self.cfg, dev = cfg, torch.device(device)
        self.device = dev
        self.logit_scale = sd["logit_scale"].float().cpu()      # a host scalar: read by every scoring call (no device -> host copy there)
        v, t = "vision_model", "text_model"
        act = "quick_gelu" if getattr(cfg, "act", "gelu") == "quick_gelu" else "gelu"
        self.v_enc = _EncoderX3(pack_clip_layers_x3(sd, v, cfg.v_layers, dev), cfg.v_heads, 1e-5, act)
        self.t_enc = _EncoderX3(pack_clip_layers_x3(sd, t, cfg.t_layers, dev), cfg.t_heads, 1e-5, act, causal=True)
        pw = sd[f"{v}.embeddings.patch_embedding.weight"].float()
        D = pw.shape[0]
        flat = torch.zeros(D, 640, dtype=f32)
        flat[:, :588] = pw.reshape(D, -1)                       # conv [D,3,14,14] -> [D, 588 (+ zero pad to the im2col pitch)]
        self.patch_w3 = _w3(flat, dev)
        pos = sd[f"{v}.embeddings.position_embedding.weight"].float()
        self.v_pos = _v(pos, dev)
        self.v_cls = _v(sd[f"{v}.embeddings.class_embedding"].float() + pos[0], dev)
        self.pre_ln = (_v(sd[f"{v}.pre_layrnorm.weight"], dev), _v(sd[f"{v}.pre_layrnorm.bias"], dev))
        self.post_ln = (_v(sd[f"{v}.post_layernorm.weight"], dev), _v(sd[f"{v}.post_layernorm.bias"], dev))
        self.v_proj3 = _w3(sd["visual_projection.weight"], dev)
        self.tok_emb = _v(sd[f"{t}.embeddings.token_embedding.weight"], dev)
        self.t_pos = _v(sd[f"{t}.embeddings.position_embedding.weight"], dev)
        self.final_ln = (_v(sd[f"{t}.final_layer_norm.weight"], dev), _v(sd[f"{t}.final_layer_norm.bias"], dev))
        self.t_proj3 = _w3(sd["text_projection.weight"], dev)
        self._id_ln = {}

    def _ln_f32(self, x, wb):
        """LayerNorm returning f32 rows: the split rows' hi + lo halves ARE the f32 value to 2^-16 -- but the residual stream
        wants the unsplit value, so this LayerNorm runs as (split -> identity product): cheap at the two places it is used."""
        D = x.shape[1]
        eye3 = ops.cached(self._id_ln, D, lambda: ops.split_x3(torch.eye(D, dtype=f32, device=x.device), 1))
        return linear_x3(ops.layernorm_x3(x, wb[0], wb[1], 1e-5), eye3)

    @torch.no_grad()
    def image_features_from_patches3(self, patches3, B):
        cfg = self.cfg
        P = (cfg.image_size // cfg.patch) ** 2
        S, D = P + 1, cfg.v_hidden
        emb = linear_x3(patches3, self.patch_w3).view(B, P, D)
        x = torch.empty(B, S, D, dtype=f32, device=emb.device)
        x[:, 0] = self.v_cls                                     # (class embedding + its position)
        x[:, 1:] = emb + self.v_pos[1:][None]
        x = self._ln_f32(x.view(B * S, D), self.pre_ln)
        x = self.v_enc(x, B, S)
        pooled = ops.layernorm_x3(x.view(B, S, D)[:, 0].contiguous(), self.post_ln[0], self.post_ln[1], 1e-5)
        return linear_x3(pooled, self.v_proj3)

    @torch.no_grad()
    def get_image_features(self, images=None, pixel_patches3=None):
        """images: [B,3,H,W] in [0,1] (device); the PIL-exact resize + normalise is fused in, pixels stay f32."""
        if pixel_patches3 is None:
            pixel_patches3 = preprocess.clip_patches(images, self.cfg.image_size, x3=True)
        B = pixel_patches3.shape[0] // ((self.cfg.image_size // self.cfg.patch) ** 2)
        return self.image_features_from_patches3(pixel_patches3, B)

    @torch.no_grad()
    def get_text_features(self, input_ids):
        cfg = self.cfg
        B, S = input_ids.shape
        D = cfg.t_hidden
        ids = input_ids.to(self.device)
        x = (self.tok_emb[ids] + self.t_pos[:S][None]).reshape(B * S, D).contiguous()
        x = self.t_enc(x, B, S)
        # transformers' CLIPTextTransformer pooling: configs with the legacy eos_token_id = 2 (the released PickScore_v1 / laion CLIP-H
        # config.json) pool at argmax(input_ids) -- the eos of the original CLIP vocabulary is its largest id; otherwise at the first eos
        eos = ids.int().argmax(dim=-1) if cfg.eos_token_id == 2 else (ids == cfg.eos_token_id).int().argmax(dim=-1)
        pooled = x.view(B, S, D)[torch.arange(B, device=self.device), eos].contiguous()
        return linear_x3(ops.layernorm_x3(pooled, self.final_ln[0], self.final_ln[1], 1e-5), self.t_proj3)


class DinoV2X3:
    """vit.DinoV2's surface (timm vit_base_patch14_dinov2 forward_features) in the fp32-equivalent arithmetic: the tower the
    reference runs in fp32 for image_similarity_score (adv_grpo/rewards.py:147-203).  Features come back in f32."""
    x3 = True

    def __init__(self, sd, cfg, device="cuda"):
        self.cfg, dev = cfg, torch.device(device)
        self.device = dev
        self.num_features = cfg.hidden
        layers = []
        for i in range(cfg.layers):
            p = f"blocks.{i}"
            layers.append({
                "ln1.w": _v(sd[f"{p}.norm1.weight"], dev), "ln1.b": _v(sd[f"{p}.norm1.bias"], dev),
                "ln2.w": _v(sd[f"{p}.norm2.weight"], dev), "ln2.b": _v(sd[f"{p}.norm2.bias"], dev),
                "qkv.w3": _w3(sd[f"{p}.attn.qkv.weight"], dev), "qkv.b": _v(sd[f"{p}.attn.qkv.bias"], dev),
                "out.w3": _w3(sd[f"{p}.attn.proj.weight"], dev), "out.b": _v(sd[f"{p}.attn.proj.bias"], dev),
                "fc1.w3": _w3(sd[f"{p}.mlp.fc1.weight"], dev), "fc1.b": _v(sd[f"{p}.mlp.fc1.bias"], dev),
                "fc2.w3": _w3(sd[f"{p}.mlp.fc2.weight"], dev), "fc2.b": _v(sd[f"{p}.mlp.fc2.bias"], dev),
                "ls1": _v(sd[f"{p}.ls1.gamma"], dev).view(1, -1), "ls2": _v(sd[f"{p}.ls2.gamma"], dev).view(1, -1)})
        self.enc = _EncoderX3(layers, cfg.heads, 1e-6, "gelu")
        pw = sd["patch_embed.proj.weight"].float()
        flat = torch.zeros(pw.shape[0], 640, dtype=f32)
        flat[:, :588] = pw.reshape(pw.shape[0], -1)
        self.patch_w3 = _w3(flat, dev)
        self.patch_b = _v(sd["patch_embed.proj.bias"], dev)
        self.pos = _v(sd["pos_embed"][0], dev)
        self.cls = _v(sd["cls_token"][0, 0].float() + sd["pos_embed"][0, 0].float(), dev)
        self.norm = (_v(sd["norm.weight"], dev), _v(sd["norm.bias"], dev))
        self._eye = {}

    def eval(self):
        return self

    @torch.no_grad()
    def forward_features(self, images=None, pixel_patches3=None):
        """images: [B,3,H,W] in [0,1] (f32); bicubic-518 + ImageNet normalise fused in, in f32.  -> [B, 1+P, D] f32."""
        cfg = self.cfg
        if pixel_patches3 is None:
            pixel_patches3 = preprocess.dino_patches(images.float(), cfg.image_size, x3=True)
        P = (cfg.image_size // cfg.patch) ** 2
        B = pixel_patches3.shape[0] // P
        S, D = P + 1, cfg.hidden
        emb = ops.add_rows_f32(linear_x3(pixel_patches3, self.patch_w3), None, self.patch_b).view(B, P, D)
        x = torch.empty(B, S, D, dtype=f32, device=emb.device)
        x[:, 0] = self.cls
        x[:, 1:] = emb + self.pos[1:][None]
        x = self.enc(x.view(B * S, D), B, S)
        eye3 = ops.cached(self._eye, D, lambda: ops.split_x3(torch.eye(D, dtype=f32, device=x.device), 1))
        return linear_x3(ops.layernorm_x3(x, self.norm[0], self.norm[1], 1e-6), eye3).view(B, S, D)


def pickscore_scores_f32(image_embs, text_embs, logit_scale):
    """pickscore_scorer.py:40-52 in f32: exp(logit_scale) * <t / |t|, i / |i|> / 26 per pair."""
    i = image_embs / image_embs.norm(dim=-1, keepdim=True)
    t = text_embs / text_embs.norm(dim=-1, keepdim=True)
    return torch.as_tensor(logit_scale, device=i.device).exp() * (t * i).sum(-1) / 26.0
