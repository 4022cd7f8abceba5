"""ratio == 1 at update 0?  Old log-probs of a rollout vs compute_log_prob of the same samples with unchanged weights."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from adv_grpo_amd import synthetic
from adv_grpo_amd.config.experiments import get_config
from adv_grpo_amd.mmdit_train import SD3TransformerLoRA
from adv_grpo_amd.model_configs import ClipConfig, MMDiTConfig, VaeConfig
from adv_grpo_amd.pickscore_scorer import PickScoreScorer
from adv_grpo_amd.pipeline import SD3Pipeline
from adv_grpo_amd.trainer import SyntheticData, Trainer
from adv_grpo_amd.vae import AutoencoderKLDecoder
from adv_grpo_amd.diffusers_patch.sd3_sde_with_logprob import sde_step_cfg
full = len(sys.argv) > 1 and sys.argv[1] == "full"
cfg = get_config("pickscore_cotrain_sd3_fast", gpu_number=1)
cfg.sample.num_batches_per_epoch = 1; cfg.train_d = False
if full:
    cfg.sample.num_image_per_prompt = 8
    mcfg = MMDiTConfig(); ccfg = ClipConfig(); data_kw = dict(resolution=cfg.resolution)
else:
    cfg.resolution = 256; cfg.sample.num_steps = 4; cfg.sample.num_image_per_prompt = cfg.sample.mini_num_image_per_prompt = 2
    mcfg = MMDiTConfig(num_layers=2, num_heads=4, joint_attention_dim=256, pooled_projection_dim=128, pos_embed_max_size=96, dual_attention_layers=(0,))
    ccfg = ClipConfig(v_layers=2, t_layers=2); data_kw = dict(n_prompts=100, n_tokens=21, ctx_dim=256, pooled_dim=128, resolution=256)
with synthetic.on_device("cuda"):
    tr = SD3TransformerLoRA(synthetic.mmdit_weights(mcfg, 1234), mcfg, "cuda", seed=cfg.seed)
    vae = AutoencoderKLDecoder(synthetic.vae_decoder_weights(VaeConfig(), 4321), VaeConfig(), "cuda", mode="bf16")
    scorer = PickScoreScorer("cuda", model_sd=synthetic.clip_weights(ccfg, 777), clip_cfg=ccfg)
trainer = Trainer(cfg, SD3Pipeline(tr, vae, "cuda"), SyntheticData(device="cuda", **data_kw), scorer, None, 0, 1)
s = trainer.sample_epoch()
G = cfg.sample.mini_num_image_per_prompt
neg_pe, neg_ppe = trainer.data.neg
embeds = torch.cat([neg_pe.repeat(G, 1, 1), s["prompt_embeds"][:G]]); pooled = torch.cat([neg_ppe.repeat(G, 1), s["pooled_prompt_embeds"][:G]])
print("first_step_index", s["first_step_index"], "timesteps", s["timesteps"][0].tolist(), "sigmas", trainer.pipe.scheduler.sigmas.tolist())
for j in range(cfg.sample.train_num_steps):
    x = s["latents"][:G, j].contiguous(); nxt = s["next_latents"][:G, j].contiguous(); ts = s["timesteps"][:G, j]
    v, ctx = tr.forward_train(torch.cat([x, x]), torch.cat([ts, ts]).float(), embeds, pooled)
    (vi,) = tr(torch.cat([x, x]), torch.cat([ts, ts]).float(), embeds, pooled)
    print(f"j={j}: training forward == inference forward: {torch.equal(v, vi)}  max diff {(v.float() - vi.float()).abs().max().item():.3e}")
    for name, vv in (("train", v), ("infer", vi)):
        _, _, lp, _, _ = sde_step_cfg(trainer.pipe.scheduler, vv[:G].contiguous(), vv[G:].contiguous(), cfg.sample.guidance_scale, None, x,
                                      cfg.sample.noise_level, prev_sample=nxt, want_mean=False, step_index=s["first_step_index"][0] + j)
        print(f"   {name}: new lp {lp.tolist()[:4]} old lp {s['log_probs'][:G, j].tolist()[:4]} max |diff| {(lp - s['log_probs'][:G, j]).abs().max().item():.3e}")
from adv_grpo_amd import g_step
s["advantages"] = torch.randn(G, cfg.sample.train_num_steps, device="cuda")
sub = {k: s[k][:G] for k in ("latents", "next_latents", "timesteps", "log_probs", "advantages", "prompt_embeds", "pooled_prompt_embeds")}
for rep in range(2):
    for j in range(cfg.sample.train_num_steps):
        info = g_step.micro_step(tr, trainer.pipe.scheduler, sub, j, embeds, pooled, sub["log_probs"][:, j], sub["advantages"][:, j],
                                 guidance_scale=cfg.sample.guidance_scale, noise_level=cfg.sample.noise_level, adv_clip_max=cfg.train.adv_clip_max,
                                 clip_range=cfg.train.clip_range, loss_scale=0.5, step_index=s["first_step_index"][0] + j)
        print(f"micro_step rep {rep} j={j}: approx_kl {float(info['approx_kl']):.3e} clipfrac {float(info['clipfrac']):.3f} loss {float(info['loss']):.4e} "
              f"max |lp - old| {(info['log_prob'] - sub['log_probs'][:, j]).abs().max().item():.3e} clip_range {cfg.train.clip_range}")
