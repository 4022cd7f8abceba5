"""The PickScore discriminator's criterion behind the reference's own call: ``CLIPCriterion(CLIPCriterionConfig())(model, batch)``.

Mirror of adv_grpo/pick_score_training.py:76-87 (``CLIPCriterionConfig``) and :89-224 (``CLIPCriterion``: ``get_features`` -> ``calc_loss``
-> ``forward(model, batch)``), as called by ``train_pickscore`` (scripts/train_sd3_fast_pickscore.py:151-183):

    loss = criterion(scorer.model, batch); optimizer.zero_grad(); loss.backward(); optimizer.step(); loss.item()

Here ``model`` is the scorer's ``vit.CLIPModel`` (``PickScoreScorer.model``) with a trainable view attached -- ``ClipLastLayerTrainable(model)``
/ ``ClipLayersTrainable(model, tune_layer)`` (d_step_pickscore.py), which is what the ``requires_grad_`` selection of TP:1016-1020 becomes
on flat parameter vectors -- or that view itself.  The forward (frozen layers, trainable layers, both towers' features, L2 norms, the 2-way
cross entropy on the diagonal text -> image logits with the batch's labels) and the explicit backward are one chain of HIP launches
(``advgrpo_clip_pair_loss_labels`` is the criterion; include/advgrpo.h); there is no autograd graph.  To keep the caller's
``zero_grad(); loss.backward()`` order meaningful, the chain writes the gradient into a staging vector and the returned
``CriterionLoss.backward()`` adds it to the trainable's accumulator.  A model without a trainable view runs forward only; ``backward()`` on
that loss raises.

The batch keys are the reference's (configurable through the config's ``*_column_name`` fields):
  input_ids [B, 77]; pixels_0 / pixels_1: CLIPProcessor ``pixel_values`` [B, 3, 224, 224] (normalised, any float dtype) or ready patch rows
  [B * 256, 640] (preprocess.pil_patches); label_0 / label_1: 0-dim or [B]; num_examples_per_prompt: read and ignored, as in the reference
  (its weighting is commented out, pick_score_training.py:190-194).
Not on the kernels, stated instead of silently approximated: ``in_batch_negatives=True`` (the image-side cross entropy over all texts) and
``is_distributed=True`` (feature all-gather with autograd through it) raise NotImplementedError -- the shipped configuration uses neither
(both default False, TP:177 passes the default config); the N > 1 D-step all-reduces the gradient vector instead (trainer.py).
"""
from dataclasses import dataclass

import torch

from . import _lib
from .d_step_pickscore import ClipLastLayerTrainable


@dataclass
class CLIPCriterionConfig:
    _target_: str = "trainer.criterions.clip_criterion.CLIPCriterion"
    is_distributed: bool = False
    label_0_column_name: str = "label_0"
    label_1_column_name: str = "label_1"
    input_ids_column_name: str = "input_ids"
    pixels_0_column_name: str = "pixels_0"
    pixels_1_column_name: str = "pixels_1"
    num_examples_per_prompt_column_name: str = "num_examples_per_prompt"
    in_batch_negatives: bool = False


class CriterionLoss:
    """What ``criterion(model, batch)`` returns: the loss as a device scalar (``.item()``, ``.detach()``, ``float()``) and a
    ``backward()`` that releases the gradient computed with it into the trainable view's accumulator (once)."""

    def __init__(self, value, trainable, staged):
        self.value, self._trainable, self._staged = value, trainable, staged

    def item(self):
        return self.value.item()

    def detach(self):
        return self.value

    def __float__(self):
        return float(self.value.item())

    def backward(self):
        if self._trainable is None:
            raise _lib.AdvGrpoError("CLIPCriterion: this loss was computed on a model without a trainable view (forward only); attach "
                                    "d_step_pickscore.ClipLastLayerTrainable(model) / ClipLayersTrainable(model, tune_layer) first")
        if self._staged is None:
            raise _lib.AdvGrpoError("CLIPCriterion: backward() called twice on one loss (the gradient was already released)")
        self._trainable.grads.add_(self._staged)
        self._staged = None


def patch_rows(pixel_values, patch=14, pad_to=640):
    """CLIPProcessor ``pixel_values`` [B, 3, S, S] -> the im2col rows the patch-embedding GEMM reads: [B * (S/patch)^2, pad_to] bf16, row =
    (channel, ky, kx) of one patch (the flatten order of the Conv2d weight [D, 3, 14, 14]), zero padding to the K granule.  Index
    plumbing only: the normalised values are carried as they are (one bf16 rounding, as the bf16 tower's first cast)."""
    B, C, H, W = pixel_values.shape
    if H % patch or W % patch:
        raise ValueError(f"pixel_values {tuple(pixel_values.shape)}: height / width must be multiples of the patch size {patch}")
    gh, gw = H // patch, W // patch
    rows = pixel_values.reshape(B, C, gh, patch, gw, patch).permute(0, 2, 4, 1, 3, 5).reshape(B * gh * gw, C * patch * patch)
    out = torch.zeros(B * gh * gw, pad_to, dtype=torch.bfloat16, device=pixel_values.device)
    out[:, :rows.shape[1]] = rows.to(torch.bfloat16)
    return out


class CLIPCriterion:
    def __init__(self, cfg: CLIPCriterionConfig):
        self.cfg = cfg
        if cfg.in_batch_negatives:
            raise NotImplementedError("CLIPCriterion(in_batch_negatives=True): only the diagonal (pairwise) criterion the shipped trainer "
                                      "uses (pick_score_training.py:172-186, default False) runs on the kernels")
        if cfg.is_distributed:
            raise NotImplementedError("CLIPCriterion(is_distributed=True): features are not all-gathered here; with N > 1 ranks the "
                                      "trainer all-reduces the discriminator's gradient vector instead (trainer.py, DESIGN.md 5)")

    @staticmethod
    def _trainable_of(model):
        if isinstance(model, ClipLastLayerTrainable):
            return model, model.m
        inner = getattr(model, "module", model)                     # a DDP-style wrapper, as the reference strips it (RW:575)
        return getattr(inner, "trainable", None), inner

    def _labels(self, batch, B, device):
        out = []
        for name in (self.cfg.label_0_column_name, self.cfg.label_1_column_name):
            v = torch.as_tensor(batch[name], dtype=torch.float32).to(device).reshape(-1)
            if v.numel() not in (1, B):
                raise ValueError(f"{name}: expected a scalar or {B} labels, got {v.numel()}")
            out.append(v.expand(B).contiguous())
        return tuple(out)

    def forward(self, model, batch):
        c = self.cfg
        trainable, clip = self._trainable_of(model)
        ids = batch[c.input_ids_column_name]
        if ids.dim() == 1:                                          # the reference's .squeeze(0) on a one-prompt batch (TP:160)
            ids = ids[None]
        B = ids.shape[0]
        batch[c.num_examples_per_prompt_column_name]                # must exist, is not used (pick_score_training.py:190-194)
        P = (clip.cfg.image_size // clip.cfg.patch) ** 2

        def rows(px):
            px = px.to(clip.device)
            if px.dim() == 3:                                       # .squeeze(0) on a one-image batch (TP:161-162)
                px = px[None]
            return px if px.dim() == 2 else patch_rows(px, clip.cfg.patch)
        patches = torch.cat([rows(batch[c.pixels_0_column_name]), rows(batch[c.pixels_1_column_name])])
        if patches.shape[0] != 2 * B * P:
            raise ValueError(f"CLIPCriterion: {B} prompts need {2 * B * P} patch rows (pixels_0 then pixels_1), got {patches.shape[0]}")
        labels = self._labels(batch, B, clip.device)
        if trainable is None:
            return CriterionLoss(self._forward_only(clip, patches, ids, labels, B), None, None)
        staged = torch.zeros_like(trainable.grads)
        keep, trainable.grads = trainable.grads, staged
        try:
            loss = trainable.loss_and_grads(patches, ids, labels=labels)
        finally:
            trainable.grads = keep
        return CriterionLoss(loss, trainable, staged)

    __call__ = forward

    @staticmethod
    @torch.no_grad()
    def _forward_only(clip, patches, ids, labels, B):
        lib = _lib.load()
        e = clip.image_features_from_patches(patches, 2 * B)
        t = clip.get_text_features(ids)
        loss = torch.empty(1, dtype=torch.float32, device=e.device)
        de = torch.empty_like(e)
        _lib.check(lib.advgrpo_clip_pair_loss_labels(e.data_ptr(), t.data_ptr(), B, e.shape[1], float(clip.logit_scale.exp()),
                                                     _lib.ptr(labels[0]), _lib.ptr(labels[1]), loss.data_ptr(), de.data_ptr(),
                                                     _lib.stream_ptr()))
        return loss[0]
