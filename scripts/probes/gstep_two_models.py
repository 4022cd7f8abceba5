import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from test_gpu_trainer import _build
tr_a, ma, _ = _build("pickscore", train_d=False)
tr_b, mb, _ = _build("pickscore", train_d=False)
print("initial params equal", torch.equal(ma.params, mb.params))
samples = tr_a.sample_epoch()
samples["advantages"] = torch.randn(samples["rewards"].shape[0], tr_a.cfg.sample.train_num_steps, device="cuda")
import adv_grpo_amd.g_step as G
G0 = tr_a.cfg.sample.mini_num_image_per_prompt
neg_pe, neg_ppe = tr_a.data.neg
for name, tr, m in (("a", tr_a, ma), ("b", tr_b, mb)):
    s = {k: samples[k][:G0] for k in ("latents", "next_latents", "timesteps", "log_probs", "advantages", "prompt_embeds", "pooled_prompt_embeds")}
    embeds = torch.cat([neg_pe.repeat(G0, 1, 1), s["prompt_embeds"]]); pooled = torch.cat([neg_ppe.repeat(G0, 1), s["pooled_prompt_embeds"]])
    m.grads.zero_()
    info = G.micro_step(m, tr.pipe.scheduler, s, 0, embeds, pooled, s["log_probs"][:, 0], s["advantages"][:, 0], guidance_scale=4.5, noise_level=0.8,
                        adv_clip_max=5.0, clip_range=1e-5, loss_scale=0.5, step_index=samples["first_step_index"][0])
    torch.cuda.synchronize()
    print(name, "log_prob", info["log_prob"].tolist(), "grad norm", m.grads.norm().item())
print("grads equal after one micro-step:", torch.equal(ma.grads, mb.grads), (ma.grads - mb.grads).abs().max().item())
sq_a = ma.optimizer_step(lr=3e-4); sq_b = mb.optimizer_step(lr=3e-4)
print("sumsq equal", torch.equal(sq_a, sq_b), "params equal after step:", torch.equal(ma.params, mb.params), (ma.params - mb.params).abs().max().item())
