"""D-step (discriminator update), DINO variant, on the gfx950 kernels.

Mirror of ``train_dino`` (scripts/train_sd3_fast_dino_patch.py:156-232) and ``DINOHead`` (TD:592-603): frozen DINOv2
backbone features of the epoch's reference (real) and generated (fake) images, hinge loss on the CLS logit plus
0.3 x hinge on 64 randomly sampled patch logits, Adam(lr=d_lr, betas=(0.5, 0.999)) on the 394 241 head parameters
(TD:749-750), gradients all-reduced across ranks (the reference wraps the head in DDP, TD:749).

Everything runs on the device: the reference's tensor -> PIL -> Resize(518, BICUBIC) -> ToTensor -> Normalize round
trip (TD:135-149,166-176) is the bit-exact PIL-resample kernel with truncating uint8 quantisation; the two Linears
are MFMA GEMMs (pre-activation kept for the backward), the first Linear's weight gradient is a split-K GEMM over the
~12k feature rows, and all parameters / moments live in one flat f32 vector updated by one fused Adam launch.
"""
import torch

from . import _lib, ops, preprocess


class DinoHeadTrainable:
    """DINOHead with flat f32 master parameters [W1 (Hd x D) | b1 (Hd) | W2 (Hd) | b2 (1)]."""

    def __init__(self, state_dict=None, in_dim=768, hidden_dim=512, device="cuda", seed=0):
        self.D, self.Hd, self.device = in_dim, hidden_dim, torch.device(device)
        n = hidden_dim * in_dim + hidden_dim + hidden_dim + 1
        self.n_params = n
        self.params = torch.zeros(n, dtype=torch.float32, device=self.device)
        if state_dict is None:
            # nn.Linear's default init -- kaiming_uniform_(a = sqrt(5)) = U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for weight and
            # bias alike -- drawn from a generator seeded with `seed`, so that every rank that passes the same seed starts
            # from the same head (the reference gets that from DDP's broadcast of rank 0's parameters, TD:749)
            g = torch.Generator().manual_seed(seed)

            def uniform(shape, fan_in):
                bound = 1.0 / fan_in ** 0.5
                return (torch.rand(shape, generator=g) * 2 - 1) * bound
            state_dict = {"layers.0.weight": uniform((hidden_dim, in_dim), in_dim), "layers.0.bias": uniform((hidden_dim,), in_dim),
                          "layers.2.weight": uniform((1, hidden_dim), hidden_dim), "layers.2.bias": uniform((1,), hidden_dim)}
        o = self._offsets()
        self.params[o[0]:o[1]] = state_dict["layers.0.weight"].float().reshape(-1).to(self.device)
        self.params[o[1]:o[2]] = state_dict["layers.0.bias"].float().to(self.device)
        self.params[o[2]:o[3]] = state_dict["layers.2.weight"].float().reshape(-1).to(self.device)
        self.params[o[3]:o[4]] = state_dict["layers.2.bias"].float().to(self.device)
        self.grads = torch.zeros_like(self.params)
        self.exp_avg = torch.zeros_like(self.params)
        self.exp_avg_sq = torch.zeros_like(self.params)
        self.p16 = self.params.to(torch.bfloat16)
        self.opt_step = 0

    def _offsets(self):
        a = self.Hd * self.D
        return (0, a, a + self.Hd, a + 2 * self.Hd, a + 2 * self.Hd + 1)

    def views(self, src):
        o = self._offsets()
        return (src[o[0]:o[1]].view(self.Hd, self.D), src[o[1]:o[2]], src[o[2]:o[3]], src[o[3]:o[4]])

    def state_dict(self):
        w1, b1, w2, b2 = self.views(self.params)
        return {"layers.0.weight": w1.clone(), "layers.0.bias": b1.clone(), "layers.2.weight": w2.view(1, -1).clone(),
                "layers.2.bias": b2.clone()}

    # the scoring interface used by rewards.dino_patch_cotrain_score (vit.DinoHead API)
    @torch.no_grad()
    def patch_score(self, feats, idx, cls_weight=0.7):
        from .vit import DinoHead
        sd = {k: v for k, v in self.state_dict().items()}
        return DinoHead(sd, self.device).patch_score(feats, idx, cls_weight)

    @torch.no_grad()
    def loss_and_grads(self, feats_real, feats_fake, idx_real, idx_fake, patch_loss_weight=0.3):
        """feats_*: [B,1+N,D] bf16; idx_*: [B,n] int64.  Accumulates grads; returns (d_loss, acc) device scalars."""
        lib = _lib.load()
        B, T, D = feats_real.shape
        n = idx_real.shape[1]
        Bt = 2 * B
        R = Bt * (1 + n)
        dev = feats_real.device
        feats = torch.cat([feats_real, feats_fake]).contiguous()
        idx = torch.cat([idx_real, idx_fake]).to(torch.int64).contiguous()
        X = torch.empty(R, D, dtype=torch.bfloat16, device=dev)
        _lib.check(lib.advgrpo_gather_rows(feats.data_ptr(), idx.data_ptr(), X.data_ptr(), Bt, T, D, n, _lib.stream_ptr()))
        w1, b1, w2, b2 = self.views(self.p16)
        pre = torch.empty(R, self.Hd, dtype=torch.bfloat16, device=dev)
        h = ops.gemm_train(X, w1, bias=b1, act="gelu", aux_out=pre)
        logits = torch.empty(R, dtype=torch.float32, device=dev)
        dl = torch.empty(R, dtype=torch.float32, device=dev)
        stats = torch.empty(4, dtype=torch.float32, device=dev)
        _lib.check(lib.advgrpo_dino_head_loss(h.data_ptr(), w2.data_ptr(), b2.data_ptr(), R, self.Hd, n, B, Bt,
                                              float(patch_loss_weight), logits.data_ptr(), dl.data_ptr(), stats.data_ptr(),
                                              _lib.stream_ptr()))
        gw1, gb1, gw2, gb2 = self.views(self.grads)
        dpre = torch.empty(R, self.Hd, dtype=torch.bfloat16, device=dev)
        _lib.check(lib.advgrpo_dino_head_dpre(pre.data_ptr(), h.data_ptr(), w2.data_ptr(), dl.data_ptr(), dpre.data_ptr(),
                                              gw2.data_ptr(), gb1.data_ptr(), R, self.Hd, _lib.stream_ptr()))
        gb2 += stats[3]
        # dW1 [Hd, D] += dpre^T X  (contraction over the R rows)
        ops.gemm_train(ops.transpose(dpre), ops.transpose(X), out=gw1, splitk=max(1, min(16, R // 512)))
        d_loss = stats[0]
        acc = 0.5 * (stats[1] / B + stats[2] / B)
        return d_loss, acc

    @torch.no_grad()
    def adam_step(self, lr, betas=(0.5, 0.999), eps=1e-8):
        lib = _lib.load()
        self.opt_step += 1
        _lib.check(lib.advgrpo_adamw_step(self.params.data_ptr(), self.p16.data_ptr(), self.grads.data_ptr(),
                                          self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(), self.n_params, lr, betas[0],
                                          betas[1], eps, 0.0, self.opt_step, None, 0.0, 1.0, _lib.stream_ptr()))


def dino_train_features(scorer, images01):
    """The D-step's image path (TD:135-149,166-184): [0,1] device images -> uint8 by truncation -> PIL-exact
    Resize(518, BICUBIC) -> ToTensor/Normalize(ImageNet) -> bf16 -> frozen backbone features."""
    patches = preprocess.pil_patches(images01, scorer.cfg.image_size, preprocess.IMAGENET_MEAN, preprocess.IMAGENET_STD,
                                     trunc=True)
    return scorer.forward_features(pixel_patches=patches)


@torch.no_grad()
def train_dino(scorer, head, prompts, reference_imgs, generated_imgs, lr, n_patches=64, patch_loss_weight=0.3,
               idx_real=None, idx_fake=None, all_reduce=None):
    """One discriminator step (TD:156-232).  reference_imgs / generated_imgs: device tensors [B,3,H,W] in [0,1]
    (the trainer holds them; the reference converts to PIL and back).  Returns (d_loss, acc) as floats."""
    fr = dino_train_features(scorer, reference_imgs)
    ff = dino_train_features(scorer, generated_imgs)
    B, N = fr.shape[0], fr.shape[1] - 1
    n = min(n_patches, N)
    if idx_real is None:
        idx_real = torch.randint(0, N, (B, n), device=fr.device)
    if idx_fake is None:
        idx_fake = torch.randint(0, N, (B, n), device=fr.device)
    d_loss, acc = head.loss_and_grads(fr, ff, idx_real, idx_fake, patch_loss_weight)
    if all_reduce is not None:                       # DDP(head) semantics: average the gradients over ranks
        all_reduce(head.grads)
    head.adam_step(lr)
    return d_loss.item(), acc.item()
