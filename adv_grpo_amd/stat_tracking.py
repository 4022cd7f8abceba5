"""Per-prompt group advantages on the device (float64 kernel, bit-exact with numpy).

Mirror of adv_grpo/stat_tracking.py:12-79 (class name, update/get_stats/clear).  Group
identity is an int32 key per sample (the dataset index) instead of the prompt string the
reference rebuilds by decoding 256 token ids per sample (train_sd3_fast_pickscore.py:960-970);
strings are still accepted and mapped to dense keys on the host.
"""
import numpy as np
import torch

from . import _lib


def group_advantage(rewards, group_ids, global_std, return_stats=False):
    """rewards: device tensor [N] or [N,T], f32/f64; group_ids: device int32 [N] -> f64 [N(,T)].
    return_stats: also the device tensor [zero_std_ratio, reward_std_mean] (f64) of calculate_zero_std_ratio, computed
    by the same launch from column 0 in the rewards' own dtype (TP:195-229)."""
    lib = _lib.load()
    r = rewards.contiguous()
    N = r.shape[0]
    T = 1 if r.dim() == 1 else r.shape[1]
    out = torch.empty(r.shape, dtype=torch.float64, device=r.device)
    g = group_ids.to(device=r.device, dtype=torch.int32).contiguous()
    if not return_stats:
        _lib.check(lib.advgrpo_group_advantage(_lib.ptr(r), _lib.dtype_code(r.dtype), _lib.ptr(g), N, T,
                                               int(bool(global_std)), _lib.ptr(out), _lib.stream_ptr()))
        return out
    stats = torch.empty(2, dtype=torch.float64, device=r.device)
    _lib.check(lib.advgrpo_group_advantage_stats(_lib.ptr(r), _lib.dtype_code(r.dtype), _lib.ptr(g), N, T,
                                                 int(bool(global_std)), _lib.ptr(out), _lib.ptr(stats), _lib.stream_ptr()))
    return out, stats


def calculate_zero_std_ratio(group_ids, gathered_rewards):
    """Mirror of calculate_zero_std_ratio (scripts/train_sd3_fast_pickscore.py:195-229): share of prompt groups whose
    reward std is exactly zero and the mean of the per-group stds.  ``gathered_rewards``: a dict with key 'ori_avg' (as
    upstream) or the [N] reward tensor itself (f32 as gathered, or f64); ``group_ids`` int keys (one per prompt) or prompt
    strings.  Returns a device tensor [2] (f64): no host sync."""
    r = gathered_rewards["ori_avg"] if isinstance(gathered_rewards, dict) else gathered_rewards
    if not (isinstance(r, torch.Tensor) and r.is_cuda):
        r = torch.as_tensor(np.asarray(r)).cuda()
    if r.dtype not in (torch.float32, torch.float64):
        r = r.double()
    if not isinstance(group_ids, torch.Tensor):
        keys = list(group_ids)
        if keys and not isinstance(keys[0], (int, np.integer)):
            order = {k: i for i, k in enumerate(sorted(set(keys)))}       # np.unique order of the prompt strings
            keys = [order[k] for k in keys]
        group_ids = torch.tensor(keys, dtype=torch.int32)
    _, stats = group_advantage(r.reshape(-1), group_ids, True, return_stats=True)
    return stats


class PerPromptStatTracker:
    def __init__(self, global_std=False, device="cuda"):
        self.global_std = global_std
        self.device = device
        self.stats = {}
        self.history_prompts = set()
        self.last_group_stats = None       # device [zero_std_ratio, reward_std_mean] of the last update (TP:975-988)
        self._pending = []

    def update(self, prompts, rewards, type="grpo"):
        if type != "grpo":
            raise NotImplementedError("only type='grpo' is on the Adv-GRPO hot path (SURVEY.md 8a13)")
        if isinstance(prompts, torch.Tensor) and prompts.is_cuda:
            # gathered int keys already on the device (the trainer's path): no host round trip here; the bookkeeping that
            # get_stats() reports is derived from them lazily, when it is asked for
            ids_dev = prompts.to(torch.int32)
            self._pending.append(ids_dev)
        else:
            keys = list(prompts.tolist()) if isinstance(prompts, (np.ndarray, torch.Tensor)) else list(prompts)
            self._account(keys)
            # dense keys in np.unique (sorted) order of the prompts: the group statistics walk the groups like upstream
            uniq = {k: i for i, k in enumerate(sorted(set(keys)))}
            ids_dev = torch.tensor([uniq[k] for k in keys], dtype=torch.int32)
        if isinstance(rewards, torch.Tensor) and rewards.is_cuda:
            r = rewards
        else:
            r = torch.as_tensor(np.asarray(rewards)).to(self.device)
        if r.dtype not in (torch.float32, torch.float64):
            r = r.to(torch.float64)
        adv, self.last_group_stats = group_advantage(r, ids_dev.to(r.device), self.global_std, return_stats=True)
        return adv

    def _account(self, keys):
        for k in keys:
            self.stats[k] = self.stats.get(k, 0) + 1
            self.history_prompts.add(hash(k))

    def get_stats(self):
        for ids in self._pending:                      # the one device -> host copy, at logging time
            self._account(ids.tolist())
        self._pending = []
        avg_group_size = sum(self.stats.values()) / len(self.stats) if self.stats else 0
        return avg_group_size, len(self.history_prompts)

    def clear(self):
        self.stats = {}
        self._pending = []
