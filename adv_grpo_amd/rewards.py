"""Scorer plugin registry and aggregator (the ``adv_grpo.rewards`` surface kept for the hot path).

Mirror of adv_grpo/rewards.py:1012-1095: ``multi_score(device, {name: weight}) -> _fn(images, prompts,
metadata, scorer=None, ref_images=None, only_strict=True, head=None, ...) -> ({name: scores, ..., 'avg':
[...]}, {})`` with the same name-dispatched calling conventions (factory takes ``device`` iff it has a
parameter of that name; co-train scorers receive the backbone / head as arguments).  Registered: the six
scorers on the hot path; the remote-server / VLM scorers are out of scope and raise a
clear error.  New scorers can be added with ``register_scorer``.
"""
import inspect

import torch

from . import vit

_OUT_OF_SCOPE = ("deqa", "video_ocr", "imagereward", "qwenvl", "aesthetic", "jpeg_compressibility", "unifiedreward",
                 "geneval", "clipscore", "image_similarity_eval", "constractive_external", "discriminator",
                 "pickscore_patch", "dino_multi_cotrain", "siglip_cotrain", "siglip_image_similarity")

_PICKSCORE_FACTORY_ARGS = {}


def configure_pickscore(model_sd, clip_cfg, tokenizer=None):
    """Weights for the stand-alone ``pickscore`` scorer (the reference downloads yuvalkirstain/PickScore_v1)."""
    _PICKSCORE_FACTORY_ARGS.update(model_sd=model_sd, clip_cfg=clip_cfg, tokenizer=tokenizer)


def pickscore_score(device):
    """rewards.py:561-574: fp32 PickScorer built by the factory."""
    from .pickscore_scorer import PickScoreScorer
    if not _PICKSCORE_FACTORY_ARGS:
        raise RuntimeError("call rewards.configure_pickscore(model_sd, clip_cfg) first (no checkpoint download)")
    scorer = PickScoreScorer(dtype=torch.float32, device=device, **_PICKSCORE_FACTORY_ARGS)

    def _fn(images, prompts, metadata):
        return scorer(prompts, images), {}
    return _fn


def pickscore_cotrain_score(device):
    """rewards.py:577-589: the co-trained scorer is passed in by the trainer."""
    def _fn(scorer, images, prompts, metadata):
        return scorer(prompts, images), {}
    return _fn


def dino_patch_cotrain_score(device, n_patches=64):
    """rewards.py:375-434.  ``scorer`` = vit.DinoV2 (preprocessing fused), ``head`` = vit.DinoHead."""
    def _fn(scorer, head, images, prompts, metadata, cls_weight=0.7, idx=None):
        images = images if isinstance(images, torch.Tensor) else torch.as_tensor(images)
        if images.shape[-1] == 3:
            images = images.permute(0, 3, 1, 2)
        if images.dtype == torch.uint8 or images.max() > 1.0:
            images = images.float() / 255.0
        feats = scorer.forward_features(images=images.to(device))
        B, N = feats.shape[0], feats.shape[1] - 1
        n = min(n_patches, N)
        if idx is None:
            idx = torch.randint(0, N, (B, n), device=feats.device)
        hybrid, cls_score, patch_scores = head.patch_score(feats, idx, cls_weight)
        return hybrid, {"cls_score": cls_score, "patch_scores": patch_scores, "patch_indices": idx,
                        "cls_weight": cls_weight}
    return _fn


def dino_cotrain_score(device):
    """rewards.py:266-294: CLS embedding, L2 norm, head."""
    def _fn(scorer, head, images, prompts, metadata):
        feats = scorer.forward_features(images=images.to(device))
        idx = torch.zeros(feats.shape[0], 1, dtype=torch.int64, device=feats.device)
        _, cls_score, _ = head.patch_score(feats, idx, 1.0)
        return cls_score, {}
    return _fn


def image_similarity_score(device):
    """rewards.py:147-203 (eval): max cosine similarity of DINOv2 CLS embeddings against reference images.
    Needs a backbone: configure with rewards.configure_dino(model)."""
    def _fn(images, ref_images):
        model = _DINO.get("model")
        if model is None:
            raise RuntimeError("call rewards.configure_dino(model) first (no checkpoint download)")
        a = model.forward_features(images=images.to(device).float())[:, 0].float()
        b = model.forward_features(images=ref_images.to(device).float())[:, 0].float()
        a = a / a.norm(dim=-1, keepdim=True)
        b = b / b.norm(dim=-1, keepdim=True)
        scores = a @ b.T
        return scores.max(dim=1).values, {"pairwise": scores}
    return _fn


_DINO = {}


def configure_dino(model):
    _DINO["model"] = model


def ocr_score(device):
    """rewards.py:675-689: PaddleOCR + Levenshtein on the host; stays a host plugin (no kernel)."""
    def _fn(images, prompts, metadata):
        raise RuntimeError("ocr scorer needs paddleocr, which is not installed on this platform")
    return _fn


score_functions = {
    "ocr": ocr_score,
    "pickscore": pickscore_score,
    "image_similarity": image_similarity_score,
    "pickscore_cotrain": pickscore_cotrain_score,
    "dino_cotrain": dino_cotrain_score,
    "dino_patch_cotrain": dino_patch_cotrain_score,
}


def register_scorer(name, factory):
    score_functions[name] = factory


def multi_score(device, score_dict):
    score_fns = {}
    for name in score_dict:
        if name not in score_functions:
            if name in _OUT_OF_SCOPE:
                raise KeyError(f"scorer '{name}' is outside the accelerated hot path (SURVEY.md section 2.1 row 5); "
                               "register a host implementation with rewards.register_scorer")
            raise KeyError(name)
        fac = score_functions[name]
        score_fns[name] = fac(device) if "device" in inspect.signature(fac).parameters else fac()

    def _fn(images, prompts, metadata, scorer=None, ref_images=None, only_strict=True, head=None, fusion=None,
            layer_ids=None, temperature=0.2):
        total = []
        details = {}
        for name, weight in score_dict.items():
            if name == "image_similarity":
                scores, _ = score_fns[name](images, ref_images)
            elif name == "pickscore_cotrain":
                scores, _ = score_fns[name](scorer, images, prompts, metadata)
            elif name in ("dino_cotrain", "dino_patch_cotrain"):
                scores, _ = score_fns[name](scorer, head, images, prompts, metadata)
            else:
                scores, _ = score_fns[name](images, prompts, metadata)
            details[name] = scores
            if isinstance(scores, torch.Tensor):
                weighted = weight * scores            # stays on the device; same arithmetic as the list version
                total = weighted if isinstance(total, list) and not total else total + weighted
            else:
                weighted = [weight * s for s in scores]
                total = weighted if isinstance(total, list) and not total else [a + b for a, b in zip(total, weighted)]
        details["avg"] = total
        return details, {}
    return _fn
