"""How much does merging LoRA into bf16 weights (W_eff = bf16(W + s B A), adv_grpo_amd/mmdit_train.py refresh) change the
policy log-prob, compared with the side path y = W x + s B (A x) that PEFT runs (TP:490-511)?  k AdamW steps from B = 0 at
lr 3e-4 on real G-step gradients (full-size SD3.5-medium, 512^2, G = 8), then the fp32 oracle evaluates log_prob with
(a) the exact effective weights W + s B A (= the side path in exact arithmetic) and (b) those weights rounded to bf16 as
the product merges them.  Printed per k: the policy's own log-prob change |d_pol|, the merge error |d_q| and d_q / d_pol."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adv_grpo_amd import g_step, synthetic
from adv_grpo_amd.mmdit_train import SD3TransformerLoRA
from adv_grpo_amd.model_configs import MMDiTConfig
from adv_grpo_amd.scheduler import FlowMatchEulerDiscreteScheduler
from oracle import lora as o_lora, mmdit as o, rollout as o_roll
from oracle.scheduler import FlowMatchEulerScheduler
cfg, ocfg = MMDiTConfig(), o.MMDiTConfig()
W = {k: v.to(torch.bfloat16) for k, v in synthetic.mmdit_weights(cfg, 1234).items()}
model = SD3TransformerLoRA(W, cfg, "cuda", seed=42)
G = 8
g = torch.Generator().manual_seed(5)
sch = FlowMatchEulerDiscreteScheduler(device="cuda"); sch.set_timesteps(10)
osch = FlowMatchEulerScheduler(); osch.device = "cuda"; osch.set_timesteps(10)
x = torch.randn(G, 16, 64, 64, generator=g).to(torch.bfloat16)
nxt = (x.float() * 0.95 + 0.3 * torch.randn(G, 16, 64, 64, generator=g)).to(torch.bfloat16)
embeds = torch.randn(2 * G, 205, 4096, generator=g).to(torch.bfloat16).cuda()
pooled = torch.randn(2 * G, 2048, generator=g).to(torch.bfloat16).cuda()
adv = torch.randn(G, generator=g).cuda()
sample = {"latents": x[:, None].cuda(), "next_latents": nxt[:, None].cuda(), "timesteps": sch.timesteps[1].repeat(G)[:, None]}
kw = dict(guidance_scale=4.5, noise_level=0.8, adv_clip_max=5, clip_range=1e-5)
W32 = {k: t.float().cuda() for k, t in W.items()}

@torch.no_grad()
def oracle_lp(weights):
    tr = lambda xx, tt, cc, pp: o.mmdit_forward(weights, ocfg, xx.float(), tt, cc.float(), pp.float())
    return o_roll.compute_log_prob(tr, osch, dict(sample), 0, embeds, pooled, guidance_scale=4.5, noise_level=0.8)[1].double()

lp_base = oracle_lp(W32)
probe = g_step.micro_step(model, sch, sample, 0, embeds, pooled, torch.zeros(G, device="cuda"), adv, **kw)
model.grads.zero_()
lp0 = probe["log_prob"].clone()
print("log_prob (product, step 0)", lp0[:4].tolist(), " oracle", lp_base[:4].tolist())
for k in range(1, 21):
    info = g_step.micro_step(model, sch, sample, 0, embeds, pooled, lp0, adv, **kw)
    model.optimizer_step(lr=3e-4, weight_decay=1e-4, max_grad_norm=1.0)
    if k in (1, 5, 20):
        lora = {n: t.float() for n, t in model.lora_state_dict().items()}
        exact = o_lora.effective_weights(W32, lora)
        merged = {n: (t.to(torch.bfloat16).float() if n in exact and exact[n] is not W32.get(n) else t) for n, t in exact.items()}
        lp_e, lp_m = oracle_lp(exact), oracle_lp(merged)
        d_pol, d_q = (lp_e - lp_base), (lp_m - lp_e)
        prod = g_step.micro_step(model, sch, sample, 0, embeds, pooled, lp0, adv, **kw)["log_prob"].double()
        model.grads.zero_()
        d_prod = prod - lp0.double()
        dw = max((exact[n] - W32[n]).abs().max().item() for n in exact if exact[n] is not W32.get(n))
        print(f"k={k:2d}: max|dW| {dw:.2e}  |d_pol| mean {d_pol.abs().mean():.3e} max {d_pol.abs().max():.3e}   merge error |d_q| mean "
              f"{d_q.abs().mean():.3e} max {d_q.abs().max():.3e}   |d_q|/|d_pol| mean {(d_q.abs() / d_pol.abs()).mean():.3f}   "
              f"product's own change mean {d_prod.abs().mean():.3e}  sign agreement with exact {(torch.sign(d_prod) == torch.sign(d_pol)).float().mean():.2f}")
