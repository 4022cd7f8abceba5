"""D-step (discriminator update), PickScore / CLIP variant, on the gfx950 kernels.

Mirror of ``train_pickscore`` (scripts/train_sd3_fast_pickscore.py:151-183) with ``CLIPCriterion``
(adv_grpo/pick_score_training.py:89-224) and the trainable set of TP:1016-1020: only
``vision_model.encoder.layers[tune_layer:]`` of the CLIP ViT-H/14 scorer gets gradients; Adam(lr=d_lr,
betas=(0.5, 0.999)) (TP:658).  ``ClipLastLayerTrainable``: the shipped ``tune_layer = -1`` (last encoder layer; 19.7 M parameters);
``ClipLayersTrainable``: ``tune_layer = -k`` in general (every token row of the last k layers).

CLIP pools the CLS token after the last layer, so for the trainable layer only the CLS row of its output matters:
forward runs the frozen 31 layers with the regular encoder, then the last layer with full K/V but a single (CLS)
query per image; the backward is the same structure reversed -- per-(image, head) single-query attention backward
producing dq (CLS row), dk, dv (all 257 rows), weight gradients as split-K GEMMs over the token axis.

DEVIATION (stated in SURVEY.md 2.2 / DESIGN.md): the discriminator gradients ARE all-reduced across ranks; the
reference wraps the scorer in DDP but strips ``.module`` on the first reward call, so its ranks silently diverge.
"""
import torch

from . import _lib, ops, preprocess


class ClipLastLayerTrainable:
    NAMES = ("ln1.w", "ln1.b", "qkv.w", "qkv.b", "out.w", "out.b", "ln2.w", "ln2.b", "fc1.w", "fc1.b", "fc2.w", "fc2.b")

    def __init__(self, clip_model):
        self.m = clip_model
        self.cfg = clip_model.cfg
        self.device = clip_model.device
        L = clip_model.v_enc.layers[-1]
        self.shapes = {k: tuple(L[k].shape) for k in self.NAMES}
        self.offs, off = {}, 0
        for k in self.NAMES:
            n = L[k].numel()
            self.offs[k] = (off, off + n)
            off += n
        self.n_params = off
        self.params = torch.empty(off, dtype=torch.float32, device=self.device)
        for k in self.NAMES:
            a, b = self.offs[k]
            self.params[a:b] = L[k].float().reshape(-1)
        self.p16 = self.params.to(torch.bfloat16)
        for k in self.NAMES:                       # the scorer's last layer now aliases the flat bf16 vector
            L[k] = self.view(self.p16, k)
        self.grads = torch.zeros_like(self.params)
        self.exp_avg = torch.zeros_like(self.params)
        self.exp_avg_sq = torch.zeros_like(self.params)
        self.opt_step = 0
        self.layer = L
        clip_model.trainable = self                # what CLIPCriterion(model, batch) finds when handed scorer.model (TP:177)

    def view(self, src, k):
        a, b = self.offs[k]
        return src[a:b].view(self.shapes[k])

    # ------------------------------------------------------------------ forward/backward of the criterion
    @torch.no_grad()
    def loss_and_grads(self, pixel_patches, input_ids, labels=None):
        """pixel_patches: [2B*256, 640] patch rows of (real images, then fake images); input_ids [B,77]; labels: None for
        (label_0, label_1) = (1, 0) (TP:170-171) or a pair of device f32 [B] vectors (CLIPCriterion's batch labels).
        Accumulates grads; returns the loss (device scalar)."""
        lib = _lib.load()
        m, cfg, L = self.m, self.cfg, self.layer
        P = (cfg.image_size // cfg.patch) ** 2
        S, D, H = P + 1, cfg.v_hidden, cfg.v_heads
        hd = D // H
        Bt = pixel_patches.shape[0] // P
        B = Bt // 2
        dev = pixel_patches.device
        bf16, f32 = torch.bfloat16, torch.float32
        M = Bt * S
        # ---- frozen part: embeddings, pre-LN, layers[:-1]
        ops.cached(m._pos_cache, Bt, lambda: m.v_pos.repeat(Bt, 1).contiguous())
        x = torch.empty(M, D, dtype=bf16, device=dev)
        ops.gemm(pixel_patches, m.patch_w, out=x, seg=(P, S, 1), residual=m._pos_cache[Bt])
        x.view(Bt, S, D)[:, 0] = m.v_cls
        x = ops.layernorm_mod(x, w=m.pre_ln[0], b=m.pre_ln[1], eps=1e-5)
        frozen, m.v_enc.layers = m.v_enc.layers, m.v_enc.layers[:-1]
        try:
            x = m.v_enc(x, Bt, S)
        finally:
            m.v_enc.layers = frozen
        text = m.get_text_features(input_ids)                                       # frozen tower, [B, proj]
        # ---- trainable last layer, CLS query only
        h1 = ops.layernorm_mod(x, w=L["ln1.w"], b=L["ln1.b"], eps=1e-5)             # [M, D]
        qkv = ops.gemm(h1, L["qkv.w"], bias=L["qkv.b"])                              # [M, 3D]
        o_cls = torch.empty(Bt, D, dtype=bf16, device=dev)
        probs = torch.empty(Bt, H, S, dtype=f32, device=dev)
        _lib.check(lib.advgrpo_cls_attention_fwd(qkv.data_ptr(), o_cls.data_ptr(), probs.data_ptr(), Bt, S, H, hd,
                                                 hd ** -0.5, _lib.stream_ptr()))
        x_cls = x.view(Bt, S, D)[:, 0].contiguous()
        x1 = ops.gemm(o_cls, L["out.w"], bias=L["out.b"], residual=x_cls)            # [Bt, D]
        h2 = ops.layernorm_mod(x1, w=L["ln2.w"], b=L["ln2.b"], eps=1e-5)
        pre = torch.empty(Bt, L["fc1.w"].shape[0], dtype=bf16, device=dev)
        mid = ops.gemm_train(h2, L["fc1.w"], bias=L["fc1.b"], act="gelu", aux_out=pre)
        x2 = ops.gemm(mid, L["fc2.w"], bias=L["fc2.b"], residual=x1)
        pooled = ops.layernorm_mod(x2, w=m.post_ln[0], b=m.post_ln[1], eps=1e-5)
        e = ops.gemm(pooled, m.v_proj)                                               # [Bt, proj]
        # ---- criterion + gradient w.r.t. e
        loss = torch.empty(1, dtype=f32, device=dev)
        de = torch.empty_like(e)
        l0, l1 = labels if labels is not None else (None, None)
        _lib.check(lib.advgrpo_clip_pair_loss_labels(e.data_ptr(), text.data_ptr(), B, e.shape[1], float(m.logit_scale.exp()),
                                                     _lib.ptr(l0), _lib.ptr(l1), loss.data_ptr(), de.data_ptr(), _lib.stream_ptr()))
        # ---- backward
        g = lambda k: self.view(self.grads, k)
        T = ops.transpose
        colsum = lambda t, out: _lib.check(lib.advgrpo_colsum_bf16(t.data_ptr(), t.stride(0), t.shape[0], t.shape[1],
                                                                   out.data_ptr(), _lib.stream_ptr()))
        ln_grads = lambda xx, dy, gw, gb: _lib.check(lib.advgrpo_ln_affine_grads(
            xx.data_ptr(), xx.stride(0), dy.data_ptr(), dy.stride(0), xx.shape[0], xx.shape[1], 1e-5, gw.data_ptr(),
            gb.data_ptr(), _lib.stream_ptr()))
        ones = lambda w: (w.float() - 1.0).to(bf16).view(1, -1)                      # LN affine as a "(1 + scale)" modulation
        dpooled = ops.gemm(de, T(m.v_proj, pad_to=8))                                # de . Wproj -> [Bt, D]
        dx2 = ops.layernorm_mod_bwd(x2, dpooled, scale0=ones(m.post_ln[0]), rows_per_batch=Bt, eps=1e-5)
        # MLP
        ops.gemm_train(T(dx2), T(mid), out=g("fc2.w"), splitk=2)                     # dW2 = dx2^T mid
        colsum(dx2, g("fc2.b"))
        dpre = ops.gemm_train(dx2, T(L["fc2.w"], pad_to=8), act="dgelu", aux_in=pre)
        ops.gemm_train(T(dpre), T(h2), out=g("fc1.w"), splitk=2)
        colsum(dpre, g("fc1.b"))
        dh2 = ops.gemm(dpre, T(L["fc1.w"], pad_to=8))
        ln_grads(x1, dh2, g("ln2.w"), g("ln2.b"))
        dx1 = ops.layernorm_mod_bwd(x1, dh2, scale0=ones(L["ln2.w"]), dres=dx2, rows_per_batch=Bt, eps=1e-5)
        # attention output projection
        ops.gemm_train(T(dx1), T(o_cls), out=g("out.w"), splitk=2)
        colsum(dx1, g("out.b"))
        do_cls = ops.gemm(dx1, T(L["out.w"], pad_to=8))
        dqkv = torch.empty(M, 3 * D, dtype=bf16, device=dev)
        _lib.check(lib.advgrpo_cls_attention_bwd(qkv.data_ptr(), probs.data_ptr(), do_cls.data_ptr(), dqkv.data_ptr(), Bt, S,
                                                 H, hd, hd ** -0.5, _lib.stream_ptr()))
        ops.gemm_train(T(dqkv), T(h1), out=g("qkv.w"), splitk=16)                    # contraction over all M tokens
        colsum(dqkv, g("qkv.b"))
        dh1 = ops.gemm(dqkv, T(L["qkv.w"], pad_to=8))
        ln_grads(x, dh1, g("ln1.w"), g("ln1.b"))
        return loss[0]

    @torch.no_grad()
    def adam_step(self, lr, betas=(0.5, 0.999), eps=1e-8):
        lib = _lib.load()
        self.opt_step += 1
        _lib.check(lib.advgrpo_adamw_step(self.params.data_ptr(), self.p16.data_ptr(), self.grads.data_ptr(),
                                          self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(), self.n_params, lr, betas[0],
                                          betas[1], eps, 0.0, self.opt_step, None, 0.0, 1.0, _lib.stream_ptr()))


class ClipLayersTrainable(ClipLastLayerTrainable):
    """``tune_layer = -k`` with k > 1 (TP:1016-1020: ``vision_model.encoder.layers[tune_layer:]`` trainable): the last k encoder layers
    of the vision tower, every token row (the layers below the last one feed ALL 257 tokens into the next layer's keys and values, so
    the single-query shortcut of ClipLastLayerTrainable applies to none of them; the last layer runs through the same general code).
    No shipped PickScore config uses it (config/grpo.py:356 has -1), so this path is built for correctness, not speed: the Linears,
    their data- and weight-gradient GEMMs, LayerNorm backward and the d-GELU epilogue are the kernels of the k = 1 path on M = 2B x 257
    rows; the attention backward (head dim 80: no flash kernel) materialises the 257 x 257 probabilities per (image, head) and runs as
    batched GEMMs on head-major copies padded to the 64-deep MFMA k tile, with `advgrpo_softmax_bwd_rows` between them."""

    def __init__(self, clip_model, tune_layer):
        k = -int(tune_layer)
        if k < 1 or k > len(clip_model.v_enc.layers):
            raise ValueError(f"tune_layer must be a negative layer count within the tower, got {tune_layer}")
        self.m, self.cfg, self.device = clip_model, clip_model.cfg, clip_model.device
        self.k = k
        layers = clip_model.v_enc.layers[-k:]
        self.shapes, self.offs, off = {}, {}, 0
        for li, L in enumerate(layers):
            for n in self.NAMES:
                key = (li, n)
                self.shapes[key] = tuple(L[n].shape)
                self.offs[key] = (off, off + L[n].numel())
                off += L[n].numel()
        self.n_params = off
        self.params = torch.empty(off, dtype=torch.float32, device=self.device)
        for li, L in enumerate(layers):
            for n in self.NAMES:
                a, b = self.offs[(li, n)]
                self.params[a:b] = L[n].float().reshape(-1)
        self.p16 = self.params.to(torch.bfloat16)
        for li, L in enumerate(layers):              # the scorer's layers now alias the flat bf16 vector
            for n in self.NAMES:
                L[n] = self.view(self.p16, (li, n))
        self.grads = torch.zeros_like(self.params)
        self.exp_avg = torch.zeros_like(self.params)
        self.exp_avg_sq = torch.zeros_like(self.params)
        self.opt_step = 0
        self.layers = layers
        clip_model.trainable = self

    # ------------------------------------------------------------------ attention backward, materialised (S = 257, head dim 80)
    def _attention_bwd(self, qkv, d_o, Bt, S, H, hd):
        """qkv [Bt*S, 3D], d_o [Bt*S, D] (bf16) -> dqkv [Bt*S, 3D]."""
        lib = _lib.load()
        D = H * hd
        bf16, f32 = torch.bfloat16, torch.float32
        hp, Sp = (hd + 63) // 64 * 64, (S + 63) // 64 * 64

        def heads(t):                                 # [Bt*S, D] column slice -> head-major [Bt*H, Sp, hp], zero padded
            out = torch.zeros(Bt * H, Sp, hp, dtype=bf16, device=t.device)
            out[:, :S, :hd] = t.reshape(Bt, S, H, hd).permute(0, 2, 1, 3).reshape(Bt * H, S, hd)
            return out
        q, k, v = (heads(qkv[:, i * D:(i + 1) * D]) for i in range(3))
        do = heads(d_o)
        sc = ops.bmm_nt(q, k, out_dtype=f32, alpha=hd ** -0.5)                   # [BH, Sp, Sp] scaled scores
        dp = ops.bmm_nt(do, v, out_dtype=f32)                                    # dP = dO V^T
        p16 = torch.empty(Bt * H, Sp, Sp, dtype=bf16, device=qkv.device)
        ds16 = torch.empty_like(p16)
        _lib.check(lib.advgrpo_softmax_bwd_rows(sc.data_ptr(), dp.data_ptr(), p16.data_ptr(), ds16.data_ptr(), Bt * H * Sp, Sp, S,
                                                float(hd ** -0.5), _lib.stream_ptr()))
        tr = lambda t: t.transpose(1, 2).contiguous()
        dq = ops.bmm_nt(ds16, tr(k))                                             # dQ = dS K       [BH, Sp, hp]
        dk = ops.bmm_nt(tr(ds16), tr(q))                                         # dK = dS^T Q
        dv = ops.bmm_nt(tr(p16), tr(do))                                         # dV = P^T dO
        dqkv = torch.empty(Bt * S, 3 * D, dtype=bf16, device=qkv.device)
        for i, t in enumerate((dq, dk, dv)):
            dqkv[:, i * D:(i + 1) * D] = t[:, :S, :hd].reshape(Bt, H, S, hd).permute(0, 2, 1, 3).reshape(Bt * S, D)
        return dqkv

    @torch.no_grad()
    def loss_and_grads(self, pixel_patches, input_ids, labels=None):
        lib = _lib.load()
        m, cfg = self.m, self.cfg
        P = (cfg.image_size // cfg.patch) ** 2
        S, D, H = P + 1, cfg.v_hidden, cfg.v_heads
        hd = D // H
        Bt = pixel_patches.shape[0] // P
        B = Bt // 2
        dev = pixel_patches.device
        bf16, f32 = torch.bfloat16, torch.float32
        M = Bt * S
        ops.cached(m._pos_cache, Bt, lambda: m.v_pos.repeat(Bt, 1).contiguous())
        x = torch.empty(M, D, dtype=bf16, device=dev)
        ops.gemm(pixel_patches, m.patch_w, out=x, seg=(P, S, 1), residual=m._pos_cache[Bt])
        x.view(Bt, S, D)[:, 0] = m.v_cls
        x = ops.layernorm_mod(x, w=m.pre_ln[0], b=m.pre_ln[1], eps=1e-5)
        frozen, m.v_enc.layers = m.v_enc.layers, m.v_enc.layers[:-self.k]
        try:
            if m.v_enc.layers:
                x = m.v_enc(x, Bt, S)
        finally:
            m.v_enc.layers = frozen
        text = m.get_text_features(input_ids)
        # ---- trainable layers, every row, keeping what their backward reads
        saved = []
        for L in self.layers:
            s = {"x_in": x}
            s["h1"] = ops.layernorm_mod(x, w=L["ln1.w"], b=L["ln1.b"], eps=1e-5)
            s["qkv"] = ops.gemm(s["h1"], L["qkv.w"], bias=L["qkv.b"])
            q3 = s["qkv"].view(Bt, S, 3 * D)
            s["o"] = ops.attention(q3[:, :, :D], q3[:, :, D:2 * D], q3[:, :, 2 * D:], H).view(M, D)
            s["x1"] = ops.gemm(s["o"], L["out.w"], bias=L["out.b"], residual=x)
            s["h2"] = ops.layernorm_mod(s["x1"], w=L["ln2.w"], b=L["ln2.b"], eps=1e-5)
            s["pre"] = torch.empty(M, L["fc1.w"].shape[0], dtype=bf16, device=dev)
            s["mid"] = ops.gemm_train(s["h2"], L["fc1.w"], bias=L["fc1.b"], act="gelu", aux_out=s["pre"])
            x = ops.gemm(s["mid"], L["fc2.w"], bias=L["fc2.b"], residual=s["x1"])
            saved.append(s)
        x_cls = x.view(Bt, S, D)[:, 0].contiguous()
        pooled = ops.layernorm_mod(x_cls, w=m.post_ln[0], b=m.post_ln[1], eps=1e-5)
        e = ops.gemm(pooled, m.v_proj)
        loss = torch.empty(1, dtype=f32, device=dev)
        de = torch.empty_like(e)
        l0, l1 = labels if labels is not None else (None, None)
        _lib.check(lib.advgrpo_clip_pair_loss_labels(e.data_ptr(), text.data_ptr(), B, e.shape[1], float(m.logit_scale.exp()),
                                                     _lib.ptr(l0), _lib.ptr(l1), loss.data_ptr(), de.data_ptr(), _lib.stream_ptr()))
        # ---- backward
        T = ops.transpose
        colsum = lambda t, out: _lib.check(lib.advgrpo_colsum_bf16(t.data_ptr(), t.stride(0), t.shape[0], t.shape[1],
                                                                   out.data_ptr(), _lib.stream_ptr()))
        ln_grads = lambda xx, dy, gw, gb: _lib.check(lib.advgrpo_ln_affine_grads(
            xx.data_ptr(), xx.stride(0), dy.data_ptr(), dy.stride(0), xx.shape[0], xx.shape[1], 1e-5, gw.data_ptr(),
            gb.data_ptr(), _lib.stream_ptr()))
        ones = lambda w: (w.float() - 1.0).to(bf16).view(1, -1)
        dpooled = ops.gemm(de, T(m.v_proj, pad_to=8))
        dcls = ops.layernorm_mod_bwd(x_cls, dpooled, scale0=ones(m.post_ln[0]), rows_per_batch=Bt, eps=1e-5)
        dx = torch.zeros(M, D, dtype=bf16, device=dev)                          # only the CLS rows of the last layer's output carry gradient
        dx.view(Bt, S, D)[:, 0] = dcls
        sk = max(1, min(16, ((M + 63) // 64)))                                     # split-K slices of the token-contracted weight gradients
        for li in reversed(range(self.k)):
            L, s = self.layers[li], saved[li]
            g = lambda n: self.view(self.grads, (li, n))
            ops.gemm_train(T(dx), T(s["mid"]), out=g("fc2.w"), splitk=sk)
            colsum(dx, g("fc2.b"))
            dpre = ops.gemm_train(dx, T(L["fc2.w"], pad_to=8), act="dgelu", aux_in=s["pre"])
            ops.gemm_train(T(dpre), T(s["h2"]), out=g("fc1.w"), splitk=sk)
            colsum(dpre, g("fc1.b"))
            dh2 = ops.gemm(dpre, T(L["fc1.w"], pad_to=8))
            ln_grads(s["x1"], dh2, g("ln2.w"), g("ln2.b"))
            dx1 = ops.layernorm_mod_bwd(s["x1"], dh2, scale0=ones(L["ln2.w"]), dres=dx, rows_per_batch=M, eps=1e-5)
            ops.gemm_train(T(dx1), T(s["o"]), out=g("out.w"), splitk=sk)
            colsum(dx1, g("out.b"))
            d_o = ops.gemm(dx1, T(L["out.w"], pad_to=8))
            dqkv = self._attention_bwd(s["qkv"], d_o, Bt, S, H, hd)
            ops.gemm_train(T(dqkv), T(s["h1"]), out=g("qkv.w"), splitk=sk)
            colsum(dqkv, g("qkv.b"))
            dh1 = ops.gemm(dqkv, T(L["qkv.w"], pad_to=8))
            ln_grads(s["x_in"], dh1, g("ln1.w"), g("ln1.b"))
            if li > 0:
                dx = ops.layernorm_mod_bwd(s["x_in"], dh1, scale0=ones(L["ln1.w"]), dres=dx1, rows_per_batch=M, eps=1e-5)
        return loss[0]


@torch.no_grad()
def train_pickscore(trainable, input_ids, real_images01, fake_images01, lr, all_reduce=None):
    """One discriminator step (TP:151-183).  real/fake: device tensors [B,3,H,W] in [0,1]; the reference's
    tensor -> PIL (truncating uint8) -> CLIPProcessor round trip runs as the PIL-exact device kernels."""
    size = trainable.cfg.image_size
    px = preprocess.pil_patches(torch.cat([real_images01, fake_images01]), size, preprocess.CLIP_MEAN, preprocess.CLIP_STD,
                                trunc=True)
    loss = trainable.loss_and_grads(px, input_ids)
    if all_reduce is not None:
        all_reduce(trainable.grads)
    trainable.adam_step(lr)
    return loss.item()
