"""Shader clock under the attention kernel: s_memtime cycles / s_memrealtime (100 MHz) ticks of one wave (ATT_ABL=32 build)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from adv_grpo_amd import ops
for S in (1229, 4301):
    B, H, D = 16, 24, 64
    qkv = torch.randn(B, S, 3 * H * D, device='cuda').to(torch.bfloat16)
    q, k, v = qkv[..., :H * D], qkv[..., H * D:2 * H * D], qkv[..., 2 * H * D:]
    lse = torch.zeros(B, H, S, dtype=torch.float32, device='cuda')
    for _ in range(3):
        ops.attention(q, k, v, H, lse=lse)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.attention(q, k, v, H, lse=lse)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    print(f"   {us:.1f} us per launch = {4 * B * H * S * S * D / us / 1e6:.0f} TFLOP/s", end="  ")
    raw = lse.view(-1)[:4].view(torch.int64).tolist()
    print(S, "wave lifetime:", raw[0], "shader cycles,", raw[1], "ticks of 100 MHz ->", round(raw[0] / raw[1] * 0.1, 3), "GHz;", round(raw[1] / 100, 1), "us; cycles per tile", round(raw[0] / ((S + 63) // 64)))
