"""Rollout with per-step log-prob and training-time replay (test infrastructure).

Restates
  * pipeline_with_logprob_random  adv_grpo/diffusers_patch/sd3_pipeline_with_logprob_fast.py:453-674
  * compute_log_prob              scripts/train_sd3_fast_pickscore.py:233-267
Pinned by tests/golden/rollout_*.npz: the reference functions run in the build container
on a duck-typed pipeline object with a seeded stand-in velocity network.
"""
import random

import torch

from .scheduler import retrieve_timesteps
from .sde import cfg_combine, sde_step_with_logprob

VAE_SCALING = 1.5305
VAE_SHIFT = 0.0609


def rollout(transformer, vae_decode, scheduler, *, prompt_embeds, pooled_prompt_embeds,
            negative_prompt_embeds, negative_pooled_prompt_embeds, num_inference_steps,
            guidance_scale, height, width, noise_level, mini_num_image_per_prompt,
            train_num_steps, process_index, sample_num_steps, random_timestep,
            latents=None, noises=None, vae_dtype=torch.float32):
    """transformer(hidden_states, timestep, encoder_hidden_states, pooled_projections) -> v.
    ``latents``: initial noise [G,16,h/8,w/8] (reference draws it in prepare_latents, PF:559-568).
    ``noises``: optional list (one per step) of the epsilon the reference draws from the global
    RNG inside the SDE step (drawn on every step, also when noise_level == 0)."""
    G = mini_num_image_per_prompt
    pe = prompt_embeds.repeat(G, 1, 1)                                       # PF:551-554
    ppe = pooled_prompt_embeds.repeat(G, 1)
    npe = negative_prompt_embeds.repeat(G, 1, 1)
    nppe = negative_pooled_prompt_embeds.repeat(G, 1)
    if latents is None:
        latents = torch.randn(G, 16, height // 8, width // 8).to(pe.dtype)
    latents = latents.to(pe.dtype)                                           # PF:564
    timesteps, _ = retrieve_timesteps(scheduler, num_inference_steps)        # PF:574
    random.seed(process_index)                                               # PF:585
    if random_timestep is None:
        random_timestep = random.randint(0, sample_num_steps // 2)
    do_cfg = guidance_scale > 1
    all_latents, all_log_probs, all_timesteps = [], [], []
    if do_cfg:
        tem_pe = torch.cat([npe, pe], dim=0)                                 # PF:598-599
        tem_ppe = torch.cat([nppe, ppe], dim=0)
    else:
        tem_pe, tem_ppe = pe, ppe
    for i, t in enumerate(timesteps):
        if i == random_timestep:                                             # PF:606-623
            cur = noise_level
            all_latents.append(latents)
        elif random_timestep < i < random_timestep + train_num_steps:
            cur = noise_level
        else:
            cur = 0
        inp = torch.cat([latents] * 2) if do_cfg else latents                # PF:625
        ts = t.expand(inp.shape[0])
        v = transformer(inp, ts, tem_pe, tem_ppe)                            # PF:630-637
        if do_cfg:
            vu, vt = v.chunk(2)
            v = cfg_combine(vu, vt, guidance_scale)                          # PF:640-642
        dtype = latents.dtype
        latents, log_prob, _, _ = sde_step_with_logprob(                     # PF:646-652
            scheduler, v.float(), t.unsqueeze(0), latents.float(), noise_level=cur,
            noise=None if noises is None else noises[i])
        if latents.dtype != dtype:
            latents = latents.to(dtype)                                      # PF:654-655
        if random_timestep <= i < random_timestep + train_num_steps:         # PF:657-660
            all_latents.append(latents)
            all_log_probs.append(log_prob)
            all_timesteps.append(t.repeat(len(latents)))
    z = (latents / VAE_SCALING) + VAE_SHIFT                                  # PF:667
    image = vae_decode(z.to(vae_dtype))                                      # PF:668-669
    image = (image / 2 + 0.5).clamp(0, 1)                                    # PF:670 postprocess("pt")
    return image, all_latents, all_log_probs, all_timesteps


def compute_log_prob(transformer, scheduler, sample, j, embeds, pooled_embeds, *,
                     guidance_scale, noise_level, cfg=True):
    """train_sd3_fast_pickscore.py:233-267."""
    if cfg:
        v = transformer(torch.cat([sample["latents"][:, j]] * 2),
                        torch.cat([sample["timesteps"][:, j]] * 2), embeds, pooled_embeds)
        vu, vt = v.chunk(2)
        v = cfg_combine(vu, vt, guidance_scale)
    else:
        v = transformer(sample["latents"][:, j], sample["timesteps"][:, j], embeds, pooled_embeds)
    return sde_step_with_logprob(scheduler, v.float(), sample["timesteps"][:, j],
                                 sample["latents"][:, j].float(),
                                 prev_sample=sample["next_latents"][:, j].float(),
                                 noise_level=noise_level)
