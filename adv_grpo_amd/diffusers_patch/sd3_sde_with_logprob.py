"""Flow-CPS SDE step with per-sample log-prob on the fused gfx950 kernel.

Drop-in for adv_grpo/diffusers_patch/sd3_sde_with_logprob.py:77-139 (same name, argument
meaning and return tuple).  ``sde_step_cfg`` is the fused entry the rollout uses: it also
folds the CFG combine and the cast back to the latent dtype
(sd3_pipeline_with_logprob_fast.py:640-655) into the same HBM pass.
"""
import math

import torch

from .. import _lib


def _sigma_tables(scheduler, timestep, B, device):
    """Per-sample (sigma, sigma_prev) device views + stride without a device sync."""
    ts = timestep.tolist() if isinstance(timestep, torch.Tensor) else list(timestep)
    idx = [scheduler.index_for_timestep(t) for t in ts]
    if len(set(idx)) == 1:
        i = idx[0]
        return scheduler.sigmas[i:i + 1], scheduler.sigmas[i + 1:i + 2], 0
    assert len(idx) == B
    ii = torch.tensor(idx, device=device)
    return scheduler.sigmas[ii].contiguous(), scheduler.sigmas[ii + 1].contiguous(), 1


def sde_step_cfg(scheduler, v_uncond, v_text, guidance_scale, timestep, sample, noise_level=0.7,
                 prev_sample=None, noise=None, seed=None, offset=0, out_dtype=None, want_mean=True,
                 step_index=None, sigmas=None):
    """Returns (prev_sample_f32_or_None, prev_sample_cast_or_None, log_prob, prev_sample_mean_or_None, std_dev_t).

    v_text=None => no CFG.  Exactly one of prev_sample (replay), noise (injected epsilon) or seed
    (in-kernel Philox) selects the mode."""
    lib = _lib.load()
    B = sample.shape[0]
    n = sample[0].numel()
    dev = sample.device
    if step_index is not None:
        tab = scheduler.sigmas if sigmas is None else sigmas       # `sigmas`: the caller's rollout-local table
        sig, sigp, stride = tab[step_index:step_index + 1], tab[step_index + 1:step_index + 2], 0
    else:
        sig, sigp, stride = _sigma_tables(scheduler, timestep, B, dev)
    v_uncond = v_uncond.contiguous()
    v_text = v_text.contiguous() if v_text is not None else None
    sample = sample.contiguous()
    if prev_sample is not None:
        mode, prev_sample = _lib.SDE_REPLAY, prev_sample.contiguous()
    elif noise is not None:
        mode, noise = _lib.SDE_EPS, noise.contiguous().float()
    else:
        if seed is None:
            raise ValueError("sampling mode needs `noise` or `seed`")
        mode = _lib.SDE_PHILOX
    replay = mode == _lib.SDE_REPLAY
    nxt = None if replay else torch.empty(sample.shape, dtype=torch.float32, device=dev)
    cast = None
    if not replay and out_dtype is not None and out_dtype != torch.float32:
        cast = torch.empty(sample.shape, dtype=out_dtype, device=dev)
    mean = torch.empty(sample.shape, dtype=torch.float32, device=dev) if want_mean else None
    lp = torch.empty(B, dtype=torch.float32, device=dev)
    std = torch.empty(B, dtype=torch.float32, device=dev)
    ws = torch.empty(max(1, lib.advgrpo_sde_step_workspace_bytes(B, n) // 4), dtype=torch.float32, device=dev)
    _lib.check(lib.advgrpo_sde_step(
        _lib.ptr(v_uncond), _lib.ptr(v_text), _lib.dtype_code(v_uncond.dtype), float(guidance_scale),
        _lib.ptr(sample), _lib.dtype_code(sample.dtype), _lib.ptr(sig), _lib.ptr(sigp), stride,
        float(math.sin(noise_level * math.pi / 2)), mode, _lib.ptr(noise) if mode == _lib.SDE_EPS else None,
        int(seed or 0), int(offset), _lib.ptr(prev_sample) if replay else None,
        _lib.dtype_code(prev_sample.dtype) if replay else 0, _lib.ptr(nxt), _lib.ptr(cast),
        _lib.dtype_code(cast.dtype) if cast is not None else 0, _lib.ptr(mean), _lib.ptr(lp), _lib.ptr(std),
        _lib.ptr(ws), B, n, _lib.stream_ptr()))
    return nxt, cast, lp, mean, std.view(-1, *([1] * (sample.dim() - 1)))


def sde_step_with_logprob(scheduler, model_output, timestep, sample, noise_level=0.7, prev_sample=None,
                          generator=None, noise=None):
    """Reference signature (sd3_sde_with_logprob.py:77-85); returns
    (prev_sample, log_prob, prev_sample_mean, std_dev_t), all float32.
    ``generator``: a torch.Generator whose next int64 seeds the in-kernel Philox stream."""
    seed = None
    if prev_sample is None and noise is None:
        if generator is not None:
            seed = int(torch.randint(0, 2 ** 62, (1,), generator=generator).item())
        else:
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())
    nxt, _, lp, mean, std = sde_step_cfg(scheduler, model_output, None, 1.0, timestep, sample, noise_level,
                                         prev_sample=prev_sample, noise=noise, seed=seed)
    if prev_sample is not None:
        nxt = prev_sample.float()
    return nxt, lp, mean, std


sde_step_with_logprob_new = sde_step_with_logprob
