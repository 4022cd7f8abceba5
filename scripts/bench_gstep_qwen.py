"""One G-step micro-batch of BASELINE config 5 at full size: Qwen-Image MMDiT (60 blocks, 24 x 128), 1024^2, CFG batch 16 (G = 8),
128 text tokens: forward with one checkpoint per block + backward with per-block recomputation.  `python scripts/bench_gstep_qwen.py [layers] [fp8]`"""
import sys, time
import torch
sys.path.insert(0, ".")
from adv_grpo_amd import g_step, synthetic
from adv_grpo_amd.model_configs import QwenMMDiTConfig
from adv_grpo_amd.qwen_mmdit import flops_per_sample_forward
from adv_grpo_amd.qwen_mmdit_train import QwenImageTransformerLoRA
from adv_grpo_amd.scheduler import FlowMatchEulerDiscreteScheduler

L = int(sys.argv[1]) if len(sys.argv) > 1 else 60
FP8 = len(sys.argv) > 2 and sys.argv[2] == "fp8"
dev = "cuda"
cfg = QwenMMDiTConfig(num_layers=L)
with synthetic.on_device(dev):
    model = QwenImageTransformerLoRA(synthetic.qwen_mmdit_weights(cfg, 4242, dtype=torch.bfloat16), cfg, dev)
if FP8:
    model.enable_fp8()
print(f"model + transposes + optimiser state: {torch.cuda.memory_allocated() / 2**30:.1f} GiB", flush=True)
G, Nt = 8, 128
sch = FlowMatchEulerDiscreteScheduler(device=dev); sch.set_timesteps(10)
g = torch.Generator(device=dev).manual_seed(1)
x = torch.randn(G, 16, 128, 128, device=dev, generator=g).to(torch.bfloat16)
nxt = (x.float() * 0.95 + 0.3 * torch.randn(G, 16, 128, 128, device=dev, generator=g)).to(torch.bfloat16)
sample = {"latents": x[:, None], "next_latents": nxt[:, None], "timesteps": sch.timesteps[1].repeat(G)[:, None]}
embeds = torch.randn(2 * G, Nt, cfg.joint_attention_dim, device=dev, generator=g).to(torch.bfloat16)
adv = torch.randn(G, device=dev, generator=g)
kw = dict(guidance_scale=4.5, noise_level=0.8, adv_clip_max=5, clip_range=1e-4, step_index=1)
probe = g_step.micro_step(model, sch, sample, 0, embeds, None, torch.zeros(G, device=dev), adv, **kw)
torch.cuda.synchronize()
torch.cuda.reset_peak_memory_stats()
n = 2
t0 = time.perf_counter()
for _ in range(n):
    g_step.micro_step(model, sch, sample, 0, embeds, None, probe["log_prob"], adv, **kw)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
f = 2 * G * flops_per_sample_forward(cfg, 4096, Nt) / 1e12
print(f"micro-step (CFG batch {2 * G}, {L} blocks, {'fp8 replay' if FP8 else 'bf16'} Linears): {dt * 1e3:.0f} ms; peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
print(f"  1 forward + 1 data gradient = {2 * f:.0f} TFLOP credited (recomputation and adapter gradients not counted): "
      f"{2 * f / dt:.0f} TFLOP/s = {2 * f / dt / 2500:.3f} of the bf16 peak")
t0 = time.perf_counter()
model.optimizer_step()
torch.cuda.synchronize()
print(f"clip + AdamW + re-merge of {model.n_params / 1e6:.0f} M adapter parameters: {(time.perf_counter() - t0) * 1e3:.0f} ms")
