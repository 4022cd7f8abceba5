"""SD3 rollout with per-step log-prob on the gfx950 kernels.

Drop-in for ``pipeline_with_logprob_random``
(adv_grpo/diffusers_patch/sd3_pipeline_with_logprob_fast.py:453-674): same keyword arguments, same return
tuple ``(image, all_latents, all_log_probs, all_timesteps)``, same cast order (SDE step in f32, latents stored
in the prompt-embedding dtype, log-prob from the pre-cast sample) and the same RNG call pattern (one epsilon
draw per step, also when noise_level == 0).  ``self`` is a pipeline object exposing ``transformer``,
``scheduler``, ``vae`` (see adv_grpo_amd/pipeline.py).

What changes underneath: per step ONE transformer call on the CFG batch, then ONE fused kernel for CFG
combine + SDE step + log-prob + cast (instead of ~12 torch kernels, a randn and two host syncs), epsilon from
an in-kernel Philox stream (or injected via ``noises=`` for parity tests), no ``index_for_timestep`` sync.
"""
import random
import threading

import torch

from ..scheduler import retrieve_timesteps
from .sd3_sde_with_logprob import sde_step_cfg

_CFG_SIDES_LOCK = threading.Lock()


@torch.no_grad()
def pipeline_with_logprob_random(self, prompt=None, prompt_2=None, prompt_3=None, height=None, width=None,
                                 num_inference_steps=28, mini_num_image_per_prompt=1, sigmas=None, guidance_scale=7.0,
                                 negative_prompt=None, negative_prompt_2=None, negative_prompt_3=None, generator=None,
                                 latents=None, prompt_embeds=None, negative_prompt_embeds=None,
                                 pooled_prompt_embeds=None, negative_pooled_prompt_embeds=None, output_type="pt",
                                 joint_attention_kwargs=None, clip_skip=None,
                                 callback_on_step_end_tensor_inputs=("latents",), max_sequence_length=256,
                                 skip_layer_guidance_scale=2.8, noise_level=0.7, train_num_steps=1, process_index=0,
                                 sample_num_steps=10, random_timestep=None, noises=None, seed=None):
    if prompt_embeds is None:
        raise ValueError("prompt strings need text encoders, which are outside the accelerated path "
                         "(SURVEY.md 8a2 / f3): pass prompt_embeds / pooled_prompt_embeds")
    height = height or self.default_sample_size * self.vae_scale_factor
    width = width or self.default_sample_size * self.vae_scale_factor
    self._guidance_scale = guidance_scale
    device = self._execution_device
    G = mini_num_image_per_prompt
    pe = prompt_embeds.to(device).repeat(G, 1, 1)                              # PF:551-554
    ppe = pooled_prompt_embeds.to(device).repeat(G, 1)
    do_cfg = guidance_scale > 1
    if do_cfg:
        npe = negative_prompt_embeds.to(device).repeat(G, 1, 1)
        nppe = negative_pooled_prompt_embeds.to(device).repeat(G, 1)
        tem_pe = torch.cat([npe, pe], dim=0)                                   # PF:598-599
        tem_ppe = torch.cat([nppe, ppe], dim=0)
    else:
        tem_pe, tem_ppe = pe, ppe
    dtype = prompt_embeds.dtype
    B = pe.shape[0]
    if latents is None and generator is not None and seed is None:
        # the eval loop's path (TP:298-299,319): initial latents from the caller's torch generator exactly as diffusers'
        # prepare_latents / randn_tensor draws them (a CPU generator draws on the CPU in the embedding dtype, then moves)
        gdev = generator.device if isinstance(generator, torch.Generator) else torch.device("cpu")
        shape = (B, self.transformer.config.in_channels, height // self.vae_scale_factor, width // self.vae_scale_factor)
        latents = torch.randn(shape, generator=generator, device=gdev, dtype=dtype).to(device)
    if seed is None:
        seed = int(torch.randint(0, 2 ** 62, (1,), generator=generator).item())
    if latents is None:                                                        # PF:559-568 prepare_latents
        latents = self.prepare_latents(B, self.transformer.config.in_channels, height, width, dtype, device, seed)
    latents = latents.to(device=device, dtype=dtype)
    timesteps, num_inference_steps = retrieve_timesteps(self.scheduler, num_inference_steps, device)   # PF:574
    # rollout-local view of the schedule: rollouts of other prompt groups run at the same time on other streams / host
    # threads (trainer.sample_epoch) and re-install the scheduler's tables; this call keeps the objects it started with
    sigma_table = self.scheduler.sigmas
    if random_timestep is None:                                                # PF:585-586: random.seed(process_index); randint
        random_timestep = random.Random(process_index).randint(0, sample_num_steps // 2)   # same draw, no global reseed
    self.last_random_timestep = random_timestep    # host int, per calling thread (pipeline.py): first recorded scheduler index
    all_latents, all_log_probs, all_timesteps = [], [], []
    n_per = latents[0].numel()
    # One Philox key per rollout call, disjoint counter ranges per draw: the initial latents use counters [0, ctr), the
    # epsilon of denoise step i the range [(1 + i) * ctr, (2 + i) * ctr).  (Separate keys seed + 1 + i per step made rank
    # r + 1's latents equal rank r's step-0 noise whenever callers numbered their ranks' seeds consecutively.)
    ctr = (B * n_per + 3) // 4 + 1
    # every adaLN modulation row of the rollout from ONE pass over the 1.5 GB modulation matrix (mmdit.precompute_mods:
    # bit for bit the rows the per-forward GEMM gives; rollout step 412.8 -> 409.6 ms on one box)
    pre = getattr(self.transformer, "precompute_mods", None)
    mods_all = pre(timesteps, tem_ppe) if pre is not None else None
    emb = getattr(self.transformer, "embed_context", None)
    ctx_rows = emb(tem_pe) if emb is not None else None                        # the context embedder is timestep-free as well
    for i in range(len(timesteps)):
        t = timesteps[i]
        if i == random_timestep:                                               # PF:606-623
            cur = noise_level
            all_latents.append(latents)
        elif random_timestep < i < random_timestep + train_num_steps:
            cur = noise_level
        else:
            cur = 0
        if do_cfg and getattr(self, "cfg_two_streams", False) and mods_all is not None and ctx_rows is not None:
            # optional schedule (config `sample.cfg_two_streams`, bench.py's `cfg_two_streams` leg; LABNOTES.md 6 round 5): the unconditional and the conditional half of the CFG batch are
            # independent until the combine -- two forwards of batch B on two HIP streams instead of one of batch 2 B (every row of
            # every kernel is independent of the others: the same bits)
            main = torch.cuda.current_stream(device)
            # one side stream per calling THREAD, created under a lock and kept together with the stream OBJECT it was made for (ADVICE r5:
            # a map keyed by the integer stream handle is mutated by concurrent rollout threads, and a handle re-used after its stream
            # was destroyed would alias the old entry)
            with _CFG_SIDES_LOCK:
                sides = self.__dict__.setdefault("_cfg_sides", {})
                ent = sides.get(threading.get_ident())
                if ent is None or ent[0].cuda_stream != main.cuda_stream:
                    ent = sides[threading.get_ident()] = (main, torch.cuda.Stream(device=device))
            side = ent[1]
            Nt = tem_pe.shape[1]
            ready = torch.cuda.Event()
            ready.record(main)
            side.wait_event(ready)
            with torch.cuda.stream(side):
                vt = self.transformer(hidden_states=latents, timestep=t.expand(B), encoder_hidden_states=tem_pe[B:], pooled_projections=tem_ppe[B:],
                                      return_dict=False, mods=mods_all[i][B:], context=ctx_rows[B * Nt:])[0]
            latents.record_stream(side)
            vu = self.transformer(hidden_states=latents, timestep=t.expand(B), encoder_hidden_states=tem_pe[:B], pooled_projections=tem_ppe[:B],
                                  return_dict=False, mods=mods_all[i][:B], context=ctx_rows[:B * Nt])[0]
            main.wait_stream(side)
            vt.record_stream(main)
        else:
            inp = torch.cat([latents] * 2) if do_cfg else latents                  # PF:625
            v = self.transformer(hidden_states=inp, timestep=t.expand(inp.shape[0]), encoder_hidden_states=tem_pe,
                                 pooled_projections=tem_ppe, joint_attention_kwargs=joint_attention_kwargs,
                                 return_dict=False, **({} if mods_all is None else {"mods": mods_all[i]}),
                                 **({} if ctx_rows is None else {"context": ctx_rows}))[0]                             # PF:630-637
            vu, vt = (v[:B], v[B:]) if do_cfg else (v, None)
        want_f32 = dtype == torch.float32
        nxt, cast, log_prob, _, _ = sde_step_cfg(                              # PF:640-655 fused
            self.scheduler, vu, vt, guidance_scale, None, latents, cur,
            noise=None if noises is None else noises[i], seed=seed, offset=(1 + i) * ctr,
            out_dtype=None if want_f32 else dtype, want_mean=False, step_index=i, sigmas=sigma_table)
        latents = nxt if want_f32 else cast
        if random_timestep <= i < random_timestep + train_num_steps:           # PF:657-660
            all_latents.append(latents)
            all_log_probs.append(log_prob)
            all_timesteps.append(t.repeat(B))
    if output_type == "latent":
        # diffusers' StableDiffusion3Pipeline.__call__ option (`if output_type == "latent": image = latents`) that the reference's trimmed copy
        # of the function dropped: the final latents are returned undecoded so that the caller can run `self.vae.decode_to_image` wherever
        # it schedules the reward computation (bench.py / Trainer: inside the reward future, beside the next group's rollout)
        return latents, all_latents, all_log_probs, all_timesteps
    image = self.vae.decode_to_image(latents)                                  # PF:667-670 (rescale, decode, postprocess)
    return image, all_latents, all_log_probs, all_timesteps
