"""Head-dim-128 attention backward at config 5's shape (B = 16, 24 heads, S = 4224): time per call and credited TFLOP/s."""
import sys, time
import torch
sys.path.insert(0, ".")
from adv_grpo_amd import ops
B, H, S, D = 16, 24, 4224, 128
g = torch.Generator(device="cuda").manual_seed(0)
qkv = torch.randn(B, S, 3 * H * D, device="cuda", generator=g).to(torch.bfloat16)
q, k, v = qkv[:, :, :H * D], qkv[:, :, H * D:2 * H * D], qkv[:, :, 2 * H * D:]
d_o = torch.randn(B, S, H * D, device="cuda", generator=g).to(torch.bfloat16)
lse = torch.empty(B, H, S, dtype=torch.float32, device="cuda")
o = ops.attention(q, k, v, H, lse=lse)
dqkv = torch.empty_like(qkv)
dq, dk, dv = dqkv[:, :, :H * D], dqkv[:, :, H * D:2 * H * D], dqkv[:, :, 2 * H * D:]
for _ in range(2):
    ops.attention_bwd(q, k, v, o, d_o, lse, H, dq, dk, dv)
torch.cuda.synchronize()
n = 5
t0 = time.perf_counter()
for _ in range(n):
    ops.attention_bwd(q, k, v, o, d_o, lse, H, dq, dk, dv)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
fl = 2.5 * 4.0 * S * S * D * B * H
print(f"attention_bwd d128 {B}x{H}x{S}: {dt * 1e3:.2f} ms, {fl / dt / 1e12:.0f} TFLOP/s credited (5 products) = {fl / dt / 2.5e15:.3f} of peak")
