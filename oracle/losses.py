"""Policy and discriminator losses, EMA schedule (test infrastructure).

Restates
  * GRPO clipped surrogate + diagnostics  scripts/train_sd3_fast_pickscore.py:1111-1162
  * CLIPCriterion.calc_loss (in_batch_negatives=False) adv_grpo/pick_score_training.py:118-199
  * train_dino hinge loss / accuracy      scripts/train_sd3_fast_dino_patch.py:186-230
  * DINOHead                              scripts/train_sd3_fast_dino_patch.py:592-603
  * EMAModuleWrapper decay / step         adv_grpo/ema.py:33-52
Pinned by tests/golden/losses.npz (made from the reference code).
"""
import torch
import torch.nn.functional as F


def grpo_loss(log_prob, old_log_prob, advantages, adv_clip_max, clip_range):
    """Returns (policy_loss, info dict) -- train_sd3_fast_pickscore.py:1111-1162 (beta == 0)."""
    adv = torch.clamp(advantages, -adv_clip_max, adv_clip_max)
    ratio = torch.exp(log_prob - old_log_prob)
    unclipped = -adv * ratio
    clipped = -adv * torch.clamp(ratio, 1.0 - clip_range, 1.0 + clip_range)
    policy_loss = torch.mean(torch.maximum(unclipped, clipped))
    info = {
        "approx_kl": 0.5 * torch.mean((log_prob - old_log_prob) ** 2),
        "clipfrac": torch.mean((torch.abs(ratio - 1.0) > clip_range).float()),
        "clipfrac_gt_one": torch.mean((ratio - 1.0 > clip_range).float()),
        "clipfrac_lt_one": torch.mean((1.0 - ratio > clip_range).float()),
        "policy_loss": policy_loss,
        "loss": policy_loss,
    }
    return policy_loss, info


def kl_loss(prev_sample_mean, prev_sample_mean_ref):
    """train_sd3_fast_pickscore.py:1126-1128 (the live line: no division by 2 std_dev_t^2)."""
    return torch.mean(((prev_sample_mean - prev_sample_mean_ref) ** 2).mean(dim=(1, 2, 3), keepdim=True))


def clip_pair_loss(text_features, image_0_features, image_1_features, logit_scale):
    """CLIPCriterion.calc_loss with label_0=1 (real), label_1=0 (fake), no in-batch negatives
    (pick_score_training.py:137-199): 2-way CE on the diagonal text->image logits.
    Equals mean softplus(s * (t.fake - t.real))."""
    all_img = torch.cat([image_0_features, image_1_features], dim=0)
    text_logits = logit_scale * text_features @ all_img.T
    l0, l1 = text_logits.chunk(2, dim=-1)
    idx = torch.arange(l0.shape[0])
    logits = torch.stack([l0[idx, idx], l1[idx, idx]], dim=-1)
    labels = torch.zeros(logits.shape[0], dtype=torch.long)
    return F.cross_entropy(logits, labels, reduction="none").mean()


class DinoHead(torch.nn.Module):
    """train_sd3_fast_dino_patch.py:592-603: Linear(in,512) -> GELU(exact) -> Linear(512,1)."""

    def __init__(self, in_dim=768, hidden_dim=512):
        super().__init__()
        self.layers = torch.nn.Sequential(torch.nn.Linear(in_dim, hidden_dim), torch.nn.GELU(),
                                          torch.nn.Linear(hidden_dim, 1))

    def forward(self, x):
        return self.layers(x)


def dino_hinge_loss(head, feats_real, feats_fake, idx_real, idx_fake, patch_loss_weight=0.3):
    """train_dino's loss given backbone features [B,1+N,D] and the sampled patch indices
    (train_sd3_fast_dino_patch.py:186-230).  Returns (d_loss, acc)."""
    cls_r, patch_r = feats_real[:, 0], feats_real[:, 1:]
    cls_f, patch_f = feats_fake[:, 0], feats_fake[:, 1:]
    lr = head(cls_r).squeeze(-1)
    lf = head(cls_f).squeeze(-1)
    image_loss = 0.5 * (torch.mean(F.relu(1.0 - lr)) + torch.mean(F.relu(1.0 + lf)))
    D = patch_r.shape[-1]
    sr = torch.gather(patch_r, 1, idx_real.unsqueeze(-1).expand(-1, -1, D))
    sf = torch.gather(patch_f, 1, idx_fake.unsqueeze(-1).expand(-1, -1, D))
    plr = head(sr).squeeze(-1)
    plf = head(sf).squeeze(-1)
    patch_loss = 0.5 * (torch.mean(F.relu(1.0 - plr)) + torch.mean(F.relu(1.0 + plf)))
    d_loss = image_loss + patch_loss_weight * patch_loss
    acc = 0.5 * ((lr > 0).float().mean().item() + (lf < 0).float().mean().item())
    return d_loss, acc


def ema_decay(optimization_step, decay=0.9):
    """ema.py:33-37."""
    return min((1 + optimization_step) / (10 + optimization_step), decay)


def ema_step(ema_params, params, optimization_step, decay=0.9, update_step_interval=8):
    """ema.py:39-52 (same-device branch), in place on ema_params."""
    one_minus = 1 - ema_decay(optimization_step, decay)
    if (optimization_step + 1) % update_step_interval == 0:
        for e, p in zip(ema_params, params):
            e.add_(one_minus * (p - e))
