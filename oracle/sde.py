"""Flow-CPS SDE step + Gaussian log-prob, CFG combine (test infrastructure).

Restates adv_grpo/diffusers_patch/sd3_sde_with_logprob.py:77-139
(``sde_step_with_logprob_new``) and the CFG combine at
sd3_pipeline_with_logprob_fast.py:640-642 / train_sd3_fast_pickscore.py:242-247.
Pinned by tests/golden/sde_step.npz (generated from the reference function).
"""
import math

import torch


def cfg_combine(noise_pred_uncond, noise_pred_text, guidance_scale):
    """u + s*(t-u), evaluated in the tensors' own dtype (bf16 in C2: three
    roundings, one per torch op) -- sd3_pipeline_with_logprob_fast.py:640-642."""
    return noise_pred_uncond + guidance_scale * (noise_pred_text - noise_pred_uncond)


def sde_step_with_logprob(scheduler, model_output, timestep, sample, noise_level=0.7,
                          prev_sample=None, generator=None, noise=None):
    """sd3_sde_with_logprob.py:77-139.  ``noise`` (not in the reference signature) lets a
    test inject the epsilon the reference would have drawn from the global RNG."""
    model_output = model_output.float()                                   # :101
    sample = sample.float()                                               # :102
    if prev_sample is not None:
        prev_sample = prev_sample.float()                                 # :104
    step_index = [scheduler.index_for_timestep(t) for t in timestep]      # :106
    prev_step_index = [s + 1 for s in step_index]                         # :107
    shape = (-1,) + (1,) * (sample.dim() - 1)
    sigma = scheduler.sigmas[step_index].view(shape)                      # :108
    sigma_prev = scheduler.sigmas[prev_step_index].view(shape)            # :109
    std_dev_t = sigma_prev * math.sin(noise_level * math.pi / 2)          # :118
    pred_original_sample = sample - sigma * model_output                  # :119
    noise_estimate = sample + model_output * (1 - sigma)                  # :120
    prev_sample_mean = (pred_original_sample * (1 - sigma_prev)
                        + noise_estimate * torch.sqrt(sigma_prev ** 2 - std_dev_t ** 2))  # :121
    if prev_sample is None:                                               # :124-131
        if noise is None:
            noise = torch.randn(model_output.shape, generator=generator,
                                dtype=model_output.dtype, device=model_output.device)
        prev_sample = prev_sample_mean + std_dev_t * noise
    log_prob = -((prev_sample.detach() - prev_sample_mean) ** 2)          # :134
    log_prob = log_prob.mean(dim=tuple(range(1, log_prob.ndim)))          # :137
    return prev_sample, log_prob, prev_sample_mean, std_dev_t
