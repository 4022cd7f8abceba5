"""One G-step micro-batch: compute_log_prob with gradient -> clipped GRPO loss -> backward into the LoRA grads.

Mirror of the inner training loop body, scripts/train_sd3_fast_pickscore.py:1102-1165 (beta == 0 in every shipped config;
beta > 0 adds the KL term of TP:1105-1108,1126-1130 against the adapter-free transformer): compute_log_prob (TP:233-267) = transformer on the CFG batch + CFG combine + SDE step in replay mode;
loss block TP:1111-1130; accelerator.backward(loss) TP:1165.  The backward is explicit:
d loss/d log_prob (grpo_loss kernel) -> d/d v_uncond, d/d v_text (sde_step_bwd kernel) -> MMDiT backward."""
import math

import torch

from . import _lib, losses
from .diffusers_patch.sd3_sde_with_logprob import sde_step_cfg


@torch.no_grad()
def micro_step(model, scheduler, sample, j, embeds, pooled_embeds, old_log_prob, advantages, *, guidance_scale,
               noise_level, adv_clip_max, clip_range, loss_scale=1.0, step_index=None, beta=0.0):
    """sample: dict with latents / next_latents [G,T,16,h,w], timesteps [G,T].  embeds / pooled: CFG-concatenated
    (negative first, TP:1084-1091).  Accumulates into model.grads; returns the diagnostics of TP:1132-1162.
    step_index: scheduler index of timestep j, carried on the host by the caller (the rollout knows it); without it the
    index is looked up from the timestep value as the reference does (scheduler.index_for_timestep, SDE:106-110), which
    costs a device -> host copy per micro-step.
    beta > 0: a second, gradient-free forward with the adapters disabled gives prev_sample_mean_ref;
    loss = policy_loss + beta * mean((prev_sample_mean - prev_sample_mean_ref)^2) and "kl_loss" joins the diagnostics."""
    lib = _lib.load()
    x = sample["latents"][:, j].contiguous()
    nxt = sample["next_latents"][:, j].contiguous()
    ts = sample["timesteps"][:, j]
    G = x.shape[0]
    v, ctx = model.forward_train(torch.cat([x, x]), torch.cat([ts, ts]).float(), embeds, pooled_embeds)
    vu, vt = v[:G].contiguous(), v[G:].contiguous()
    if step_index is None:
        step_index = scheduler.index_for_timestep(float(ts[0]))
    _, _, log_prob, _, _ = sde_step_cfg(scheduler, vu, vt, guidance_scale, None, x, noise_level, prev_sample=nxt,
                                        want_mean=False, step_index=step_index)
    mean_ref = None
    if beta > 0:
        v_ref = model.forward_reference(torch.cat([x, x]), torch.cat([ts, ts]).float(), embeds, pooled_embeds)
        _, _, _, mean_ref, _ = sde_step_cfg(scheduler, v_ref[:G].contiguous(), v_ref[G:].contiguous(), guidance_scale, None, x,
                                            noise_level, prev_sample=nxt, want_mean=True, step_index=step_index)
    scal, dlp = losses.grpo_loss(log_prob, old_log_prob, advantages, adv_clip_max, clip_range)
    if loss_scale != 1.0:
        dlp = dlp * loss_scale
    gu, gt = torch.empty_like(vu), torch.empty_like(vt)
    n = vu[0].numel()
    sig, sigp = scheduler.sigmas[step_index:step_index + 1], scheduler.sigmas[step_index + 1:step_index + 2]
    common = (vu.data_ptr(), vt.data_ptr(), _lib.dtype_code(vu.dtype), float(guidance_scale), x.data_ptr(),
              _lib.dtype_code(x.dtype), sig.data_ptr(), sigp.data_ptr(), 0, float(math.sin(noise_level * math.pi / 2)),
              nxt.data_ptr(), _lib.dtype_code(nxt.dtype), dlp.data_ptr())
    kl = None
    if mean_ref is None:
        _lib.check(lib.advgrpo_sde_step_bwd(*common, gu.data_ptr(), gt.data_ptr(), G, n, _lib.stream_ptr()))
    else:
        kl = torch.empty(G, dtype=torch.float32, device=x.device)
        ws = torch.empty(max(1, lib.advgrpo_sde_step_workspace_bytes(G, n) // 4), dtype=torch.float32, device=x.device)
        _lib.check(lib.advgrpo_sde_step_bwd_kl(*common, mean_ref.data_ptr(), float(beta) * float(loss_scale), gu.data_ptr(),
                                               gt.data_ptr(), kl.data_ptr(), ws.data_ptr(), G, n, _lib.stream_ptr()))
    model.backward(ctx, torch.cat([gu, gt]))
    info = {"log_prob": log_prob, "scalars": scal, **{k: scal[i] for i, k in enumerate(losses.INFO_KEYS)}}
    if kl is not None:                                   # TP:1126-1130,1158-1160
        info["kl_loss"] = kl.mean()
        info["loss"] = info["policy_loss"] + beta * info["kl_loss"]
    return info
