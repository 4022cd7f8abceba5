"""The path's exchange steps over torch.distributed (backend "nccl" = RCCL over xGMI on MI355X; "gloo" in the
CPU tests).  One packed all-gather replaces the reference's per-key gathers (rewards 'avg' [N_loc,T], 'ori_avg',
each scorer key) and the [N_loc,256] int64 prompt-id gather (scripts/train_sd3_fast_pickscore.py:926-966): rewards
and the int32 group key travel together as [N_loc, T+1] f32 (the key is exact in f32 below 2^24 prompts).
The result is a rank-major concatenation, which is what the un-gather slice at TP:995-999 assumes."""
import torch
import torch.distributed as dist


def world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def gather_rewards(rewards, group_ids):
    """rewards [N_loc,T] f32, group_ids [N_loc] int -> (rewards [n*N_loc,T] f32, group_ids [n*N_loc] int32)."""
    if not group_ids.is_cuda:                      # no device sync on the hot path
        assert int(group_ids.max()) < (1 << 24)
    n = world()
    if n == 1:
        return rewards, group_ids.to(torch.int32)
    T = rewards.shape[1]
    packed = torch.cat([rewards.float(), group_ids.view(-1, 1).float()], dim=1).contiguous()
    out = torch.empty(n * packed.shape[0], T + 1, dtype=torch.float32, device=packed.device)
    dist.all_gather_into_tensor(out, packed)
    return out[:, :T].contiguous(), out[:, T].to(torch.int32)


def ungather(advantages, num_processes=None, process_index=None):
    """TP:995-999: keep this rank's rows of a rank-major gathered tensor."""
    n = world() if num_processes is None else num_processes
    r = rank() if process_index is None else process_index
    return advantages.reshape(n, -1, advantages.shape[-1])[r]


def all_mean(t):
    """accelerator.gather(x).mean() (TP:1008-1011): identical on every rank, so the D/G gate cannot diverge."""
    n = world()
    if n == 1:
        return t.float().mean()
    s = torch.stack([t.float().sum(), torch.tensor(float(t.numel()), device=t.device)])
    dist.all_reduce(s)
    return s[0] / s[1]


def reduce_mean(info):
    """accelerator.reduce(info, reduction="mean") (TP:1179-1183): the training diagnostics of an optimizer step (loss, approx_kl, clipfrac,
    policy_loss, ...) averaged over ranks before rank 0 logs them, so the logged curve is the node's and not rank 0's.  The values (device
    scalars or host floats) travel as ONE f32 vector in key order; every rank gets the same dict back."""
    n = world()
    if n == 1 or not info:
        return dict(info)
    keys = sorted(info)
    dev = next((v.device for v in info.values() if isinstance(v, torch.Tensor)), torch.device("cpu"))
    vec = torch.stack([(info[k].detach().float().reshape(()) if isinstance(info[k], torch.Tensor) else torch.tensor(float(info[k]))).to(dev)
                       for k in keys])
    dist.all_reduce(vec)
    vec /= n
    return {k: vec[i] for i, k in enumerate(keys)}


def average_gradients(flat):
    """The update half's exchange: all-reduce (sum) of a flat gradient vector, then / world -- what DeepSpeed / DDP do to
    the LoRA gradients inside ``accelerator.backward`` (TP:1165) and DDP to the DINO head (TD:749).  In place; every rank
    ends with the same vector, so the optimizer steps that follow stay identical without a parameter broadcast."""
    n = world()
    if n > 1:
        dist.all_reduce(flat)
        flat /= n
    return flat


def broadcast_state(tensors, src=0):
    """Rank `src`'s trainable state on every rank (DDP / DeepSpeed broadcast at construction, TD:749, TP:554-561)."""
    if world() > 1:
        for t in tensors:
            dist.broadcast(t, src=src)
