"""D-step (discriminator update), PickScore / CLIP variant, on the gfx950 kernels.

Mirror of ``train_pickscore`` (scripts/train_sd3_fast_pickscore.py:151-183) with ``CLIPCriterion``
(adv_grpo/pick_score_training.py:89-224) and the trainable set of TP:1016-1020: only
``vision_model.encoder.layers[tune_layer:]`` of the CLIP ViT-H/14 scorer gets gradients; Adam(lr=d_lr,
betas=(0.5, 0.999)) (TP:658).  Built for the shipped ``tune_layer = -1`` (last encoder layer; 19.7 M parameters).

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

    def view(self, src, k):
        a, b = self.offs[k]
        return src[a:b].view(self.shapes[k])

    # ------------------------------------------------------------------ forward/backward of the criterion
    @torch.no_grad()
    def loss_and_grads(self, pixel_patches, input_ids):
        """pixel_patches: [2B*256, 640] patch rows of (real images, then fake images); input_ids [B,77].
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
        _lib.check(lib.advgrpo_clip_pair_loss(e.data_ptr(), text.data_ptr(), B, e.shape[1], float(m.logit_scale.exp()),
                                              loss.data_ptr(), de.data_ptr(), _lib.stream_ptr()))
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
