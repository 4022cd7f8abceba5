"""Timing of one G-step micro-batch (CFG batch 16, SD3.5-medium, 512^2) and of the optimizer step."""
import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adv_grpo_amd import synthetic, g_step
from adv_grpo_amd.mmdit_train import SD3TransformerLoRA
from adv_grpo_amd.model_configs import MMDiTConfig
from adv_grpo_amd.scheduler import FlowMatchEulerDiscreteScheduler
cfg = MMDiTConfig()
with synthetic.on_device("cuda"):
    model = SD3TransformerLoRA(synthetic.mmdit_weights(cfg, 1234), cfg, "cuda")
# "fp8": the product's fp8 mode.  What-if runs (results are wrong, timing only): "nowgrad" skips the adapter-gradient kernels,
# "serial" runs them in the chain
mode = sys.argv[1] if len(sys.argv) > 1 else ""
if mode == "nowgrad":
    model._lora_wgrad_group = lambda *a, **k: None
if mode == "serial":
    model.overlap_wgrad = False
if mode == "fp8":                     # block Linears of the forward on e4m3 operands (enable_fp8); backward unchanged (bf16)
    model.enable_fp8()
G = 8
sch = FlowMatchEulerDiscreteScheduler(device="cuda"); sch.set_timesteps(10)
x = torch.randn(G, 1, 16, 64, 64, device="cuda").to(torch.bfloat16)
nxt = (x.float() * 0.95 + 0.3 * torch.randn_like(x.float())).to(torch.bfloat16)
sample = {"latents": x, "next_latents": nxt, "timesteps": sch.timesteps[1].repeat(G)[:, None]}
embeds = torch.randn(2 * G, 205, 4096, device="cuda").to(torch.bfloat16)
pooled = torch.randn(2 * G, 2048, device="cuda").to(torch.bfloat16)
old = torch.full((G,), -0.75, device="cuda"); adv = torch.randn(G, device="cuda")
kw = dict(guidance_scale=4.5, noise_level=0.8, adv_clip_max=5, clip_range=1e-5)
for _ in range(2): info = g_step.micro_step(model, sch, sample, 0, embeds, pooled, old, adv, **kw)
torch.cuda.synchronize(); t0 = time.time(); n = 3
for _ in range(n): info = g_step.micro_step(model, sch, sample, 0, embeds, pooled, old, adv, **kw)
torch.cuda.synchronize(); dt = (time.time() - t0) / n
# SURVEY 8d's count for a frozen base model: 1 forward + ~1 data-gradient pass of the Linears (no weight-gradient GEMM; the rank-32
# adapter gradients are negligible) and forward + 2.5x forward for attention: 2 x 1.913 + 3.5 x 0.306 = 4.90 TFLOP per sample
flop = (2 * 1.913 + 3.5 * 0.306) * 16
print(f"[{mode or 'product'}] G-step micro-batch (fwd+bwd, batch 16): {dt*1e3:.1f} ms  -> {flop/dt/1e3:.3f} PFLOP/s = {flop/dt/1e3/2.5:.3f} of the bf16 MFMA peak "
      f"({flop:.1f} TFLOP per micro-step: 1 fwd + 1 dgrad of the Linears, fwd + bwd of attention)")
print("log_prob", info["log_prob"][:3].tolist(), "peak mem GB", torch.cuda.max_memory_allocated() / 2**30)
torch.cuda.synchronize(); t0 = time.time()
model.optimizer_step()
torch.cuda.synchronize(); print(f"optimizer step + re-merge: {(time.time()-t0)*1e3:.1f} ms")
