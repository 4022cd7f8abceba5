"""Attention backward (delta + dQ + dK/dV kernels) timing at the G-step's shapes.  Usage: bench_attention_bwd.py [shape index]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adv_grpo_amd import ops
def bench(B, H, S, iters=10):
    D = 64
    qkv = torch.randn(B, S, 3 * H * D, device="cuda").to(torch.bfloat16)
    q, k, v = qkv[..., :H * D], qkv[..., H * D:2 * H * D], qkv[..., 2 * H * D:]
    out = torch.empty(B, S, H * D, dtype=torch.bfloat16, device="cuda")
    lse = torch.empty(B, H, S, dtype=torch.float32, device="cuda")
    ops.attention(q, k, v, H, out=out, lse=lse)
    do = torch.randn_like(out)
    g = torch.empty_like(qkv)
    dq, dk, dv = g[..., :H * D], g[..., H * D:2 * H * D], g[..., 2 * H * D:]
    for _ in range(2): ops.attention_bwd(q, k, v, out, do, lse, H, dq, dk, dv)
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(iters): ops.attention_bwd(q, k, v, out, do, lse, H, dq, dk, dv)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / iters
    fl = 2.5 * 4 * B * H * S * S * D           # SURVEY's count: 2.5 x the forward
    print(f"B={B} H={H} S={S}: {ms * 1e3:.1f} us  {fl / ms / 1e9:.1f} TFLOP/s credited ({fl * 1.4 / ms / 1e9:.1f} executed: 7 of 5 GEMM units)")
shapes = [(16, 24, 1229), (16, 24, 1024), (8, 38, 4301)]
for i, sh in enumerate(shapes):
    if len(sys.argv) < 2 or int(sys.argv[1]) == i: bench(*sh)
