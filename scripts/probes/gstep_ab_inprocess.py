"""Same-process A/B of G-step variants that are class switches (SD3TransformerLoRA.fuse_gates, .overlap_wgrad): the model is built once,
the variants alternate, 10 micro-steps per sample, 5 rounds.  (Separate processes per variant, 3 micro-steps each as scripts/bench_gstep.py
times them, disagreed with themselves by 3 % depending on which variant ran first.)   Usage: gstep_ab_inprocess.py [attr=fuse_gates]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from adv_grpo_amd import synthetic, g_step
from adv_grpo_amd.mmdit_train import SD3TransformerLoRA
from adv_grpo_amd.model_configs import MMDiTConfig
from adv_grpo_amd.scheduler import FlowMatchEulerDiscreteScheduler
attr = sys.argv[1] if len(sys.argv) > 1 else "fuse_gates"
cfg = MMDiTConfig()
with synthetic.on_device("cuda"):
    model = SD3TransformerLoRA(synthetic.mmdit_weights(cfg, 1234), cfg, "cuda")
G = 8
sch = FlowMatchEulerDiscreteScheduler(device="cuda"); sch.set_timesteps(10)
x = torch.randn(G, 1, 16, 64, 64, device="cuda").to(torch.bfloat16)
nxt = (x.float() * 0.95 + 0.3 * torch.randn_like(x.float())).to(torch.bfloat16)
sample = {"latents": x, "next_latents": nxt, "timesteps": sch.timesteps[1].repeat(G)[:, None]}
embeds = torch.randn(2 * G, 205, 4096, device="cuda").to(torch.bfloat16)
pooled = torch.randn(2 * G, 2048, device="cuda").to(torch.bfloat16)
old = torch.full((G,), -0.75, device="cuda"); adv = torch.randn(G, device="cuda")
kw = dict(guidance_scale=4.5, noise_level=0.8, adv_clip_max=5, clip_range=1e-5)


def run(n):
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n):
        if attr == "merge_one_launch":           # this switch is the optimizer step's: clip + AdamW + re-merge
            model.grads.normal_()
            model.optimizer_step()
            continue
        g_step.micro_step(model, sch, sample, 0, embeds, pooled, old, adv, **kw)
    torch.cuda.synchronize()
    return (time.time() - t0) / n * 1e3


run(5)
res = {True: [], False: []}
for r in range(5):
    for v in (True, False) if r % 2 == 0 else (False, True):
        setattr(model, attr, v)
        run(2)
        res[v].append(run(10))
for v in (True, False):
    print(f"{attr}={v}: " + " ".join(f"{t:.2f}" for t in res[v]) + f"   median {sorted(res[v])[2]:.2f} ms")
