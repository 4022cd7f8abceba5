"""Per-prompt group advantages on the device (float64 kernel, bit-exact with numpy).

Mirror of adv_grpo/stat_tracking.py:12-79 (class name, update/get_stats/clear).  Group
identity is an int32 key per sample (the dataset index) instead of the prompt string the
reference rebuilds by decoding 256 token ids per sample (train_sd3_fast_pickscore.py:960-970);
strings are still accepted and mapped to dense keys on the host.
"""
import numpy as np
import torch

from . import _lib


def group_advantage(rewards, group_ids, global_std):
    """rewards: device tensor [N] or [N,T], f32/f64; group_ids: device int32 [N] -> f64 [N(,T)]."""
    lib = _lib.load()
    r = rewards.contiguous()
    N = r.shape[0]
    T = 1 if r.dim() == 1 else r.shape[1]
    out = torch.empty(r.shape, dtype=torch.float64, device=r.device)
    g = group_ids.to(device=r.device, dtype=torch.int32).contiguous()
    _lib.check(lib.advgrpo_group_advantage(_lib.ptr(r), _lib.dtype_code(r.dtype), _lib.ptr(g), N, T,
                                           int(bool(global_std)), _lib.ptr(out), _lib.stream_ptr()))
    return out


class PerPromptStatTracker:
    def __init__(self, global_std=False, device="cuda"):
        self.global_std = global_std
        self.device = device
        self.stats = {}
        self.history_prompts = set()

    def update(self, prompts, rewards, type="grpo"):
        if type != "grpo":
            raise NotImplementedError("only type='grpo' is on the Adv-GRPO hot path (SURVEY.md 8a13)")
        keys = list(prompts.tolist()) if isinstance(prompts, (np.ndarray, torch.Tensor)) else list(prompts)
        uniq = {}
        ids = np.empty(len(keys), dtype=np.int32)
        for i, k in enumerate(keys):
            ids[i] = uniq.setdefault(k, len(uniq))
            self.history_prompts.add(hash(k))
        for k, gi in uniq.items():
            self.stats.setdefault(k, 0)
            self.stats[k] += int((ids == gi).sum())
        if isinstance(rewards, torch.Tensor) and rewards.is_cuda:
            r = rewards
        else:
            r = torch.as_tensor(np.asarray(rewards)).to(self.device)
        if r.dtype not in (torch.float32, torch.float64):
            r = r.to(torch.float64)
        adv = group_advantage(r, torch.from_numpy(ids).to(r.device), self.global_std)
        return adv

    def get_stats(self):
        avg_group_size = sum(self.stats.values()) / len(self.stats) if self.stats else 0
        return avg_group_size, len(self.history_prompts)

    def clear(self):
        self.stats = {}
