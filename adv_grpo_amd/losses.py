"""GRPO clipped surrogate on the device (forward + backward + diagnostics in one launch).

Mirror of the inline block scripts/train_sd3_fast_pickscore.py:1111-1162 (beta == 0)."""
import torch

from . import _lib

INFO_KEYS = ("loss", "approx_kl", "clipfrac", "clipfrac_gt_one", "clipfrac_lt_one", "policy_loss")


def grpo_loss(log_prob, old_log_prob, advantages, adv_clip_max, clip_range, want_grad=True):
    """Returns (scalars[6] device tensor in INFO_KEYS order, d loss / d log_prob or None)."""
    lib = _lib.load()
    B = log_prob.numel()
    dev = log_prob.device
    scal = torch.empty(6, dtype=torch.float32, device=dev)
    grad = torch.empty(B, dtype=torch.float32, device=dev) if want_grad else None
    # contiguous f32 copies are bound to names for the length of the call: a temporary made inside the argument list is freed
    # as soon as its pointer has been taken, and the NEXT temporary of the same size gets its block -- `old_log_prob[:, j]` (a
    # strided view) then pointed at the advantages and the kernel differentiated exp(log_prob - advantage)
    lp32, old32, adv32 = log_prob.float().contiguous(), old_log_prob.float().contiguous(), advantages.float().contiguous()
    _lib.check(lib.advgrpo_grpo_loss(_lib.ptr(lp32), _lib.ptr(old32), _lib.ptr(adv32), B, float(adv_clip_max),
                                     float(clip_range), _lib.ptr(scal), _lib.ptr(grad), _lib.stream_ptr()))
    return scal, grad
