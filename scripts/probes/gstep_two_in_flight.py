"""What would two G-step micro-steps in flight be worth (two host threads, one HIP stream each, a gradient buffer each)?  Upper bound measured with TWO
model instances so that nothing is shared: n micro-steps on one model, then n on each of two models at the same time; same process.
(The rollouts gain 4 - 5 % from two prompt groups in flight)"""
import os, sys, threading, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from adv_grpo_amd import synthetic, g_step
from adv_grpo_amd.mmdit_train import SD3TransformerLoRA
from adv_grpo_amd.model_configs import MMDiTConfig
from adv_grpo_amd.scheduler import FlowMatchEulerDiscreteScheduler
cfg = MMDiTConfig()
with synthetic.on_device("cuda"):
    W = synthetic.mmdit_weights(cfg, 1234)
    models = [SD3TransformerLoRA(W, cfg, "cuda") for _ in range(2)]
G = 8
sch = FlowMatchEulerDiscreteScheduler(device="cuda"); sch.set_timesteps(10)
x = torch.randn(G, 1, 16, 64, 64, device="cuda").to(torch.bfloat16)
nxt = (x.float() * 0.95 + 0.3 * torch.randn_like(x.float())).to(torch.bfloat16)
sample = {"latents": x, "next_latents": nxt, "timesteps": sch.timesteps[1].repeat(G)[:, None]}
embeds = torch.randn(2 * G, 205, 4096, device="cuda").to(torch.bfloat16)
pooled = torch.randn(2 * G, 2048, device="cuda").to(torch.bfloat16)
old = torch.full((G,), -0.75, device="cuda"); adv = torch.randn(G, device="cuda")
kw = dict(guidance_scale=4.5, noise_level=0.8, adv_clip_max=5, clip_range=1e-5)
streams = [torch.cuda.Stream() for _ in range(2)]


def work(k, n):
    with torch.cuda.stream(streams[k]):
        for _ in range(n):
            g_step.micro_step(models[k], sch, sample, 0, embeds, pooled, old, adv, **kw)
        streams[k].synchronize()


def serial(n):
    torch.cuda.synchronize(); t0 = time.time()
    work(0, n)
    torch.cuda.synchronize()
    return (time.time() - t0) / n * 1e3


def both(n):
    torch.cuda.synchronize(); t0 = time.time()
    ths = [threading.Thread(target=work, args=(k, n)) for k in range(2)]
    for t in ths: t.start()
    for t in ths: t.join()
    torch.cuda.synchronize()
    return (time.time() - t0) / (2 * n) * 1e3


serial(3); both(2)
a, b = [], []
for r in range(4):
    a.append(serial(8)); b.append(both(4))
print("one micro-step at a time :", " ".join(f"{t:.2f}" for t in a), "ms per micro-step")
print("two in flight (2 models) :", " ".join(f"{t:.2f}" for t in b), "ms per micro-step")
