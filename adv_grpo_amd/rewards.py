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
# scorers a shipped experiment of config/grpo.py names although they are outside SURVEY.md 8's six: why, per name
_WHY_NOT = {
    "dino_multi_cotrain": "the multi-layer DINOv2 fusion scorer (adv_grpo/rewards.py:1032, config/grpo.py `dino_cotrain_sd3_multi_fast`, tune_layer = (11,), "
                          "temperature 2) has no trainer among the two shipped hot loops (train_sd3_fast_{pickscore,dino_patch}.py index "
                          "rewards['pickscore_cotrain' | 'dino_patch_cotrain'] only) and is not one of SURVEY.md 8's six scorers: not built",
}

class PromptBatch(list):
    """The ``prompts`` argument of the scorer plugins when both faces of a prompt are needed: a list of the prompt STRINGS
    (what text-matching host scorers such as ``ocr`` read) that also carries the CLIP ``input_ids`` [N,77] of the same
    prompts for the scorers whose tokenizer cannot run on this platform (PickScore)."""

    def __init__(self, texts, clip_ids=None):
        super().__init__(texts)
        self.clip_ids = clip_ids


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
    """rewards.py:147-203 (eval): max over the reference images of the cosine similarity of DINOv2 CLS embeddings.
    On the kernels of the path: fused preprocess + DINOv2 tower, CLS rows L2-normalised by ``gather_l2norm_rows``, the
    [N, M] similarity matrix by the MFMA GEMM (f32 out); only the row maximum is index plumbing.  The reference runs this
    tower in fp32: configure a vit_x3.DinoV2X3 backbone for that arithmetic (split-bf16 products, f32 in between); with a
    vit.DinoV2 backbone the tower is bf16-MFMA / f32-accumulate (deviations).
    Needs a backbone: configure with rewards.configure_dino(model)."""
    def _cls_rows(model, images):
        from . import _lib
        lib = _lib.load()
        images = images if isinstance(images, torch.Tensor) else torch.as_tensor(images)
        if images.shape[-1] == 3:                                               # NHWC -> NCHW (RW:165-166)
            images = images.permute(0, 3, 1, 2)
        if images.dtype == torch.uint8 or images.max() > 1.0:                   # RW:163-164
            images = images.float() / 255.0
        if getattr(model, "x3", False):                                         # fp32-equivalent tower: f32 CLS rows
            cls = model.forward_features(images=images.to(device).float())[:, 0]
            return (cls / cls.norm(dim=-1, keepdim=True)).contiguous()
        feats = model.forward_features(images=images.to(device).to(torch.bfloat16)).contiguous()   # [N, 1+P, D], final norm applied
        N, T, Dm = feats.shape
        rows = torch.empty(N, Dm, dtype=torch.bfloat16, device=feats.device)    # n = 0: the CLS row of every image only
        _lib.check(lib.advgrpo_gather_l2norm_rows(_lib.ptr(feats), None, rows.data_ptr(), N, T, Dm, 0, 0.0, _lib.stream_ptr()))
        return rows

    def _fn(images, ref_images):
        from . import ops
        model = _DINO.get("model")
        if model is None:
            raise RuntimeError("call rewards.configure_dino(model) first (no checkpoint download)")
        a, b = _cls_rows(model, images), _cls_rows(model, ref_images)
        if a.dtype == torch.float32:                                            # split-bf16 product of the f32 rows
            a, b = ops.split_x3(a, 0), ops.split_x3(b, 1)
        scores = ops.gemm(a, b, out_dtype=torch.float32)                        # [N, M] cosine similarities (RW:191)
        return scores.max(dim=1).values, {"pairwise": scores}
    return _fn


_DINO = {}


def configure_dino(model):
    _DINO["model"] = model


_OCR = {}


def configure_ocr(recognizer):
    """Text recogniser for the ``ocr`` scorer: ``uint8 [H,W,3] ndarray -> str`` (PaddleOCR is used when importable)."""
    _OCR["recognizer"] = recognizer


def ocr_score(device):
    """rewards.py:675-689: quantise to uint8 NHWC (on the device), recognise + Levenshtein on the host (ocr.OcrScorer).
    ``prompts`` must be the prompt STRINGS (the target is the quoted part, ocr.py:32)."""
    from .ocr import OcrScorer
    scorer = OcrScorer(recognizer=_OCR.get("recognizer"))

    def _fn(images, prompts, metadata):
        if isinstance(images, torch.Tensor):
            images = (images.float() * 255).round().clamp(0, 255).to(torch.uint8).permute(0, 2, 3, 1).contiguous().cpu().numpy()
        return scorer(images, prompts), {}
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
                raise KeyError(f"scorer '{name}' is outside the accelerated hot path (SURVEY.md section 2.1 row 5)"
                               + (f": {_WHY_NOT[name]}" if name in _WHY_NOT else "") +
                               "; register a host implementation with rewards.register_scorer")
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
            else:
                weighted = [weight * s for s in scores]
            if isinstance(total, list) and not total:                               # RW:1086-1087
                total = weighted
            elif isinstance(total, torch.Tensor) or isinstance(weighted, torch.Tensor):
                # RW:1088-1089 adds element by element, so a list-returning scorer (ocr) and a tensor-returning one
                # (pickscore) mix in either order: python float + f32 element = f32 element
                ref = total if isinstance(total, torch.Tensor) else weighted
                a = total if isinstance(total, torch.Tensor) else torch.as_tensor(total, dtype=torch.float64, device=ref.device).to(ref.dtype)
                b = weighted if isinstance(weighted, torch.Tensor) else torch.as_tensor(weighted, dtype=torch.float64, device=ref.device).to(ref.dtype)
                total = a + b
            else:
                total = [a + b for a, b in zip(total, weighted)]
        details["avg"] = total
        return details, {}
    return _fn
