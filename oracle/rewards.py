"""Reward aggregation and scorer epilogues (test infrastructure).

Restates
  * multi_score._fn aggregation           adv_grpo/rewards.py:1043-1093
  * dino_patch_cotrain_score._preprocess  adv_grpo/rewards.py:379-391
  * dino_patch_cotrain_score._fn epilogue adv_grpo/rewards.py:393-434
  * PickScoreScorer score epilogue        adv_grpo/pickscore_scorer.py:40-52
  * uint8 image quantisation              adv_grpo/rewards.py:567-569
Pinned by tests/golden/rewards.npz (made from the reference functions).
"""
import torch
import torch.nn.functional as F

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def weighted_sum(score_dict, per_scorer_scores):
    """rewards.py:1084-1092: python-list weighted accumulation in dict order."""
    total = []
    details = {}
    for name, weight in score_dict.items():
        scores = per_scorer_scores[name]
        details[name] = scores
        weighted = [weight * s for s in scores]
        total = weighted if not total else [a + b for a, b in zip(total, weighted)]
    details["avg"] = total
    return details


def dino_preprocess(images, cuda_semantics=False):
    """rewards.py:379-391: bicubic (no antialias) to 518, ImageNet normalise, cast bf16.

    For bf16 inputs torch-CPU rounds the four cubic weights to bf16 while torch-CUDA (what the reference
    runs on) keeps weights and accumulation in f32 and rounds only the result.  cuda_semantics=False
    reproduces the CPU op bit-exactly (this is what the golden made on the CPU pins);
    cuda_semantics=True is the GPU arithmetic the HIP kernel follows."""
    if cuda_semantics and images.dtype == torch.bfloat16:
        images = F.interpolate(images.float(), size=(518, 518), mode="bicubic", align_corners=False).to(torch.bfloat16)
    else:
        images = F.interpolate(images, size=(518, 518), mode="bicubic", align_corners=False)
    mean = torch.tensor(IMAGENET_MEAN, device=images.device)[None, :, None, None]
    std = torch.tensor(IMAGENET_STD, device=images.device)[None, :, None, None]
    return ((images - mean) / std).to(torch.bfloat16)


def dino_patch_score(feats, head, idx, cls_weight=0.7):
    """rewards.py:399-421 given backbone features [B,1+N,D] and patch indices [B,n]."""
    cls_emb = feats[:, 0, :]
    patch_emb = feats[:, 1:, :]
    D = patch_emb.shape[-1]
    sampled = torch.gather(patch_emb, 1, idx.unsqueeze(-1).expand(-1, -1, D))
    cls_emb = cls_emb / (cls_emb.norm(dim=-1, keepdim=True) + 1e-6)
    sampled = sampled / (sampled.norm(dim=-1, keepdim=True) + 1e-6)
    cls_score = head(cls_emb).squeeze(-1)
    patch_scores = head(sampled).squeeze(-1)
    hybrid = cls_weight * cls_score + (1 - cls_weight) * patch_scores.mean(dim=1)
    return hybrid, cls_score, patch_scores


def pickscore_from_embeddings(image_embs, text_embs, logit_scale_log):
    """pickscore_scorer.py:40-52: exp(logit_scale) * cos(text_i, image_i) / 26."""
    image_embs = image_embs / image_embs.norm(p=2, dim=-1, keepdim=True)
    text_embs = text_embs / text_embs.norm(p=2, dim=-1, keepdim=True)
    scores = logit_scale_log.exp() * (text_embs @ image_embs.T)
    return scores.diag() / 26


def to_uint8(images):
    """rewards.py:567: (x*255).round().clamp(0,255).to(uint8) on [N,3,H,W] in [0,1]."""
    return (images * 255).round().clamp(0, 255).to(torch.uint8)
