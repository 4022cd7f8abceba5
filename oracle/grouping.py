"""Prompt sharding, group advantages and group statistics (test infrastructure).

Restates
  * DistributedKRepeatSampler        scripts/train_sd3_fast_pickscore.py:87-129
  * PerPromptStatTracker (type grpo)  adv_grpo/stat_tracking.py:12-79
  * calculate_zero_std_ratio         scripts/train_sd3_fast_pickscore.py:195-229
  * the un-gather slice              scripts/train_sd3_fast_pickscore.py:995-999
Pinned by tests/golden/{sampler,stat_tracker}.npz|json made from the reference classes.
"""
import numpy as np
import torch


def k_repeat_indices(dataset_len, batch_size, k, num_replicas, seed, epoch):
    """One iteration of the sampler for every rank: list[num_replicas] of list[batch_size]
    (train_sd3_fast_pickscore.py:102-126)."""
    total = num_replicas * batch_size
    assert total % k == 0, f"k can not divide n*b, k{k}-num_replicas{num_replicas}-batch_size{batch_size}"
    m = total // k
    g = torch.Generator()
    g.manual_seed(seed + epoch)
    indices = torch.randperm(dataset_len, generator=g)[:m].tolist()
    repeated = [idx for idx in indices for _ in range(k)]
    shuffle = torch.randperm(len(repeated), generator=g).tolist()
    shuffled = [repeated[i] for i in shuffle]
    return [shuffled[r * batch_size:(r + 1) * batch_size] for r in range(num_replicas)]


def group_advantages(group_keys, rewards, global_std):
    """PerPromptStatTracker.update(type='grpo') for a fresh tracker (the trainer clears it
    every epoch, train_sd3_fast_pickscore.py:989) -- stat_tracking.py:18-47.

    group_keys: sequence of hashable/np-comparable keys (prompt strings in the reference,
    dataset indices in the build).  rewards: [N] or [N,T].  float64 throughout."""
    keys = np.array(group_keys)
    rewards = np.array(rewards, dtype=np.float64)
    adv = np.zeros_like(rewards)
    for key in np.unique(keys):
        sel = keys == key
        grp = rewards[sel]
        mean = np.mean(np.stack(list(grp)), axis=0, keepdims=True)
        if global_std:
            std = np.std(rewards, axis=0, keepdims=True) + 1e-4
        else:
            std = np.std(np.stack(list(grp)), axis=0, keepdims=True) + 1e-4
        adv[sel] = (grp - mean) / std
    return adv


def zero_std_ratio(group_keys, ori_avg):
    """calculate_zero_std_ratio -- train_sd3_fast_pickscore.py:195-229."""
    keys = np.array(group_keys)
    _, inverse, counts = np.unique(keys, return_inverse=True, return_counts=True)
    grouped = np.asarray(ori_avg)[np.argsort(inverse)]
    groups = np.split(grouped, np.cumsum(counts)[:-1])
    stds = np.array([np.std(g) for g in groups])
    return np.count_nonzero(stds == 0) / len(stds), stds.mean()


def ungather(advantages, num_processes, process_index):
    """train_sd3_fast_pickscore.py:995-999 (gather is a rank-major concat)."""
    a = np.asarray(advantages)
    return a.reshape(num_processes, -1, a.shape[-1])[process_index]
