"""ViT reward towers on the gfx950 kernels: CLIP ViT-H/14 (PickScore_v1) and DINOv2 ViT-B/14.

Stand in for transformers' ``CLIPModel.get_image_features / get_text_features``
(adv_grpo/pickscore_scorer.py:40-44) and timm's ``forward_features`` (adv_grpo/rewards.py:397,
scripts/train_sd3_fast_dino_patch.py:183-184).  One encoder-layer routine serves all three towers:
LayerNorm (row kernel) -> packed QKV GEMM -> fused attention reading q|k|v in place -> out-proj GEMM with
the residual (and DINOv2's LayerScale as the gate operand) in its epilogue -> LayerNorm -> MLP GEMMs with
GELU / residual epilogues.
"""
import torch

from . import _lib, ops, preprocess


def _bf(t, dev):
    return t.to(device=dev, dtype=torch.bfloat16).contiguous()


class _Encoder:
    """Pre-LN transformer encoder stack over packed weights."""

    def __init__(self, layers, heads, eps, act, causal=False):
        self.layers, self.heads, self.eps, self.act, self.causal = layers, heads, eps, act, causal

    c_stack = True             # forward through advgrpo_vit_forward (one C-ABI call for the whole stack); False: the launches one by one

    _FIELDS = ("ln1.w", "ln1.b", "qkv.w", "qkv.b", "out.w", "out.b", "ls1", "ln2.w", "ln2.b", "fc1.w", "fc1.b", "fc2.w", "fc2.b", "ls2")

    def _table(self):
        """The stack's weights as the C entry's host array of advgrpo_vit_layer, cached on the data pointers of every tensor it names (the
        D-steps swap `layers` lists and re-merge weights: any change rebuilds it; the tensors themselves are held by `layers`)."""
        p = lambda t: t.data_ptr() if t is not None else None
        key = tuple(p(L.get(n)) for L in self.layers for n in self._FIELDS)
        cache = self.__dict__.setdefault("_tables", {})
        tab = cache.get(key)
        if tab is None:
            if len(cache) > 8:
                cache.clear()
            tab = (_lib.VitLayer * len(self.layers))()
            for t, L in zip(tab, self.layers):
                for n in self._FIELDS:
                    setattr(t, n.replace(".", "_"), p(L.get(n)))
            cache[key] = tab
        return tab

    def _forward_c(self, x, B, S):
        import ctypes
        lib = _lib.load()
        D, F = x.shape[1], self.layers[0]["fc1.w"].shape[0]
        assert x.is_contiguous() and x.dtype == torch.bfloat16
        d = _lib.VitDesc()
        d.B, d.S, d.D, d.H, d.mlp, d.n_layers, d.act, d.causal, d.eps = B, S, D, self.heads, F, len(self.layers), ops.ACT[self.act], int(self.causal), self.eps
        tab = self._table()
        d.x, d.layers = x.data_ptr(), ctypes.cast(tab, ctypes.POINTER(_lib.VitLayer))
        ws = torch.empty(int(lib.advgrpo_vit_workspace_bytes(B, S, D, F)), dtype=torch.uint8, device=x.device)
        _lib.check(lib.advgrpo_vit_forward(ctypes.byref(d), ws.data_ptr(), ws.numel(), _lib.stream_ptr()))
        return x

    def __call__(self, x, B, S):
        if self.c_stack and self.layers:
            return self._forward_c(x, B, S)
        D = x.shape[1]
        H = self.heads
        for L in self.layers:
            h = ops.layernorm_mod(x, w=L["ln1.w"], b=L["ln1.b"], eps=self.eps)
            qkv = ops.gemm(h, L["qkv.w"], bias=L["qkv.b"]).view(B, S, 3 * D)
            o = ops.attention(qkv[:, :, :D], qkv[:, :, D:2 * D], qkv[:, :, 2 * D:], H, causal=self.causal)
            g1 = L.get("ls1")
            ops.gemm(o.view(B * S, D), L["out.w"], bias=L["out.b"], residual=x, out=x,
                     gate=g1, gate_rows=(B * S if g1 is not None else 0))
            h = ops.layernorm_mod(x, w=L["ln2.w"], b=L["ln2.b"], eps=self.eps)
            m = ops.gemm(h, L["fc1.w"], bias=L["fc1.b"], act=self.act)
            g2 = L.get("ls2")
            ops.gemm(m, L["fc2.w"], bias=L["fc2.b"], residual=x, out=x, gate=g2,
                     gate_rows=(B * S if g2 is not None else 0))
        return x


def pack_clip_layers(sd, pfx, n, dev):
    """transformers CLIP encoder layers -> the packed per-layer dicts _Encoder consumes (q|k|v fused)."""
    out = []
    for i in range(n):
        p = f"{pfx}.encoder.layers.{i}"
        a = f"{p}.self_attn"
        out.append({
            "ln1.w": _bf(sd[f"{p}.layer_norm1.weight"], dev), "ln1.b": _bf(sd[f"{p}.layer_norm1.bias"], dev),
            "ln2.w": _bf(sd[f"{p}.layer_norm2.weight"], dev), "ln2.b": _bf(sd[f"{p}.layer_norm2.bias"], dev),
            "qkv.w": _bf(torch.cat([sd[f"{a}.q_proj.weight"], sd[f"{a}.k_proj.weight"], sd[f"{a}.v_proj.weight"]]), dev),
            "qkv.b": _bf(torch.cat([sd[f"{a}.q_proj.bias"], sd[f"{a}.k_proj.bias"], sd[f"{a}.v_proj.bias"]]), dev),
            "out.w": _bf(sd[f"{a}.out_proj.weight"], dev), "out.b": _bf(sd[f"{a}.out_proj.bias"], dev),
            "fc1.w": _bf(sd[f"{p}.mlp.fc1.weight"], dev), "fc1.b": _bf(sd[f"{p}.mlp.fc1.bias"], dev),
            "fc2.w": _bf(sd[f"{p}.mlp.fc2.weight"], dev), "fc2.b": _bf(sd[f"{p}.mlp.fc2.bias"], dev)})
    return out


def _pad_patch_weight(w, dev):
    """conv [D,3,14,14] -> [D, 640] bf16 (588 real columns + zero pad, matching the im2col rows)."""
    D = w.shape[0]
    flat = w.reshape(D, -1)
    out = torch.zeros(D, 640, dtype=torch.float32, device=w.device)
    out[:, :flat.shape[1]] = flat
    return _bf(out, dev)


class CLIPModel:
    """PickScore_v1 / CLIP ViT-H/14 with transformers state_dict names."""

    def __init__(self, sd, cfg, device="cuda"):
        self.cfg, dev = cfg, torch.device(device)
        self.device = dev
        self.logit_scale = sd["logit_scale"].float().cpu()      # a host scalar: read by every scoring call (no device -> host copy there)

        layers = lambda pfx, n: pack_clip_layers(sd, pfx, n, dev)
        act = "gelu"
        v, t = "vision_model", "text_model"
        self.v_enc = _Encoder(layers(v, cfg.v_layers), cfg.v_heads, 1e-5, act)
        self.t_enc = _Encoder(layers(t, cfg.t_layers), cfg.t_heads, 1e-5, act, causal=True)
        self.patch_w = _pad_patch_weight(sd[f"{v}.embeddings.patch_embedding.weight"], dev)
        pos = sd[f"{v}.embeddings.position_embedding.weight"]
        self.v_pos = _bf(pos, dev)
        self.v_cls = _bf(sd[f"{v}.embeddings.class_embedding"] + pos[0], dev)
        self.pre_ln = (_bf(sd[f"{v}.pre_layrnorm.weight"], dev), _bf(sd[f"{v}.pre_layrnorm.bias"], dev))
        self.post_ln = (_bf(sd[f"{v}.post_layernorm.weight"], dev), _bf(sd[f"{v}.post_layernorm.bias"], dev))
        self.v_proj = _bf(sd["visual_projection.weight"], dev)
        self.tok_emb = _bf(sd[f"{t}.embeddings.token_embedding.weight"], dev)
        self.t_pos = _bf(sd[f"{t}.embeddings.position_embedding.weight"], dev)
        self.final_ln = (_bf(sd[f"{t}.final_layer_norm.weight"], dev), _bf(sd[f"{t}.final_layer_norm.bias"], dev))
        self.t_proj = _bf(sd["text_projection.weight"], dev)
        self._pos_cache = {}

    @torch.no_grad()
    def image_features_from_patches(self, patches, B):
        cfg = self.cfg
        P = (cfg.image_size // cfg.patch) ** 2
        S, D = P + 1, cfg.v_hidden
        ops.cached(self._pos_cache, B, lambda: self.v_pos.repeat(B, 1).contiguous())
        x = torch.empty(B * S, D, dtype=torch.bfloat16, device=patches.device)
        ops.gemm(patches, self.patch_w, out=x, seg=(P, S, 1), residual=self._pos_cache[B])
        x.view(B, S, D)[:, 0] = self.v_cls
        x = ops.layernorm_mod(x, w=self.pre_ln[0], b=self.pre_ln[1], eps=1e-5)
        x = self.v_enc(x, B, S)
        pooled = ops.layernorm_mod(x.view(B, S, D)[:, 0].contiguous(), w=self.post_ln[0], b=self.post_ln[1], eps=1e-5)
        return ops.gemm(pooled, self.v_proj)

    @torch.no_grad()
    def get_image_features(self, images=None, pixel_patches=None):
        """images: [B,3,H,W] in [0,1] (device) -- preprocessing (PIL-exact resize + normalise) is fused in."""
        if pixel_patches is None:
            pixel_patches = preprocess.clip_patches(images, self.cfg.image_size)
        B = pixel_patches.shape[0] // ((self.cfg.image_size // self.cfg.patch) ** 2)
        return self.image_features_from_patches(pixel_patches, B)

    @torch.no_grad()
    def get_text_features(self, input_ids):
        cfg = self.cfg
        B, S = input_ids.shape
        D = cfg.t_hidden
        ids = input_ids.to(self.device)
        # embedding lookup + position add: index plumbing (one gather), done with torch
        x = (self.tok_emb[ids] + self.t_pos[:S][None]).reshape(B * S, D).contiguous()
        x = self.t_enc(x, B, S)
        # transformers' CLIPTextTransformer pooling: configs with the legacy eos_token_id = 2 (the released PickScore_v1 / laion CLIP-H
        # config.json) pool at argmax(input_ids) -- the eos of the original CLIP vocabulary is its largest id; otherwise at the first eos
        eos = ids.int().argmax(dim=-1) if cfg.eos_token_id == 2 else (ids == cfg.eos_token_id).int().argmax(dim=-1)
        pooled = x.view(B, S, D)[torch.arange(B, device=self.device), eos].contiguous()
        pooled = ops.layernorm_mod(pooled, w=self.final_ln[0], b=self.final_ln[1], eps=1e-5)
        return ops.gemm(pooled, self.t_proj)


class DinoV2:
    """timm vit_base_patch14_dinov2 (forward_features) with timm state_dict names."""

    def __init__(self, sd, cfg, device="cuda"):
        self.cfg, dev = cfg, torch.device(device)
        self.device = dev
        self.num_features = cfg.hidden
        layers = []
        for i in range(cfg.layers):
            p = f"blocks.{i}"
            layers.append({
                "ln1.w": _bf(sd[f"{p}.norm1.weight"], dev), "ln1.b": _bf(sd[f"{p}.norm1.bias"], dev),
                "ln2.w": _bf(sd[f"{p}.norm2.weight"], dev), "ln2.b": _bf(sd[f"{p}.norm2.bias"], dev),
                "qkv.w": _bf(sd[f"{p}.attn.qkv.weight"], dev), "qkv.b": _bf(sd[f"{p}.attn.qkv.bias"], dev),
                "out.w": _bf(sd[f"{p}.attn.proj.weight"], dev), "out.b": _bf(sd[f"{p}.attn.proj.bias"], dev),
                "fc1.w": _bf(sd[f"{p}.mlp.fc1.weight"], dev), "fc1.b": _bf(sd[f"{p}.mlp.fc1.bias"], dev),
                "fc2.w": _bf(sd[f"{p}.mlp.fc2.weight"], dev), "fc2.b": _bf(sd[f"{p}.mlp.fc2.bias"], dev),
                "ls1": _bf(sd[f"{p}.ls1.gamma"], dev).view(1, -1), "ls2": _bf(sd[f"{p}.ls2.gamma"], dev).view(1, -1)})
        self.enc = _Encoder(layers, cfg.heads, 1e-6, "gelu")
        self.patch_w = _pad_patch_weight(sd["patch_embed.proj.weight"], dev)
        self.patch_b = _bf(sd["patch_embed.proj.bias"], dev)
        self.pos = _bf(sd["pos_embed"][0], dev)
        self.cls = _bf(sd["cls_token"][0, 0] + sd["pos_embed"][0, 0], dev)
        self.norm = (_bf(sd["norm.weight"], dev), _bf(sd["norm.bias"], dev))
        self._pos_cache = {}

    def eval(self):
        return self

    @torch.no_grad()
    def forward_features(self, images=None, pixel_patches=None):
        """images: [B,3,H,W] in [0,1]; the reference's bicubic-518 + ImageNet normalise is fused in."""
        cfg = self.cfg
        if pixel_patches is None:
            pixel_patches = preprocess.dino_patches(images, cfg.image_size)
        P = (cfg.image_size // cfg.patch) ** 2
        B = pixel_patches.shape[0] // P
        S, D = P + 1, cfg.hidden
        ops.cached(self._pos_cache, B, lambda: self.pos.repeat(B, 1).contiguous())
        x = torch.empty(B * S, D, dtype=torch.bfloat16, device=pixel_patches.device)
        ops.gemm(pixel_patches, self.patch_w, bias=self.patch_b, out=x, seg=(P, S, 1), residual=self._pos_cache[B])
        x.view(B, S, D)[:, 0] = self.cls
        x = self.enc(x, B, S)
        return ops.layernorm_mod(x, w=self.norm[0], b=self.norm[1], eps=1e-6).view(B, S, D)


class DinoHead:
    """DINOHead (train_sd3_fast_dino_patch.py:592-603) scoring epilogue on device."""

    def __init__(self, sd, device="cuda"):
        dev = torch.device(device)
        self.w1, self.b1 = _bf(sd["layers.0.weight"], dev), _bf(sd["layers.0.bias"], dev)
        self.w2, self.b2 = _bf(sd["layers.2.weight"].reshape(-1), dev), _bf(sd["layers.2.bias"], dev)

    @torch.no_grad()
    def patch_score(self, feats, idx, cls_weight=0.7):
        """feats [B,1+N,D] bf16, idx [B,n] int64 -> (hybrid[B], cls[B], patch[B,n]) f32 (rewards.py:399-421)."""
        from . import _lib
        lib = _lib.load()
        B, T, D = feats.shape
        n = idx.shape[1]
        rows = torch.empty(B * (1 + n), D, dtype=torch.bfloat16, device=feats.device)
        feats_c, idx_c = feats.contiguous(), idx.to(torch.int64).contiguous()      # (named: temporaries inside the argument list
        _lib.check(lib.advgrpo_gather_l2norm_rows(_lib.ptr(feats_c), _lib.ptr(idx_c),  #  are freed before the launch)
                                                  rows.data_ptr(), B, T, D, n, 1e-6, _lib.stream_ptr()))
        hid = ops.gemm(rows, self.w1, bias=self.b1, act="gelu")
        hyb = torch.empty(B, dtype=torch.float32, device=feats.device)
        cls = torch.empty_like(hyb)
        pat = torch.empty(B, n, dtype=torch.float32, device=feats.device)
        _lib.check(lib.advgrpo_dino_head_combine(hid.data_ptr(), self.w2.data_ptr(), self.b2.data_ptr(), B, hid.shape[1], n,
                                                 float(cls_weight), hyb.data_ptr(), cls.data_ptr(), pat.data_ptr(),
                                                 _lib.stream_ptr()))
        return hyb, cls, pat


def pickscore_scores(image_embs, text_embs, logit_scale):
    """pickscore_scorer.py:40-52 epilogue on device: [B,P] bf16 x2 -> [B] f32."""
    from . import _lib
    lib = _lib.load()
    B, P = image_embs.shape
    out = torch.empty(B, dtype=torch.float32, device=image_embs.device)
    image_c, text_c = image_embs.contiguous(), text_embs.contiguous()
    _lib.check(lib.advgrpo_pickscore_pairs(_lib.ptr(image_c), _lib.ptr(text_c), B, P,
                                           float(torch.as_tensor(logit_scale).exp()), out.data_ptr(), _lib.stream_ptr()))
    return out
