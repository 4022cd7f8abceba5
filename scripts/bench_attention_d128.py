"""Head-dim-128 attention forward at the Qwen-Image shape (config 5: B = 16 incl. CFG, 24 heads, S = 4096 + text) and others.
Usage: bench_attention_d128.py [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adv_grpo_amd import ops


def bench(B, H, S, iters=10, D=128):
    qkv = torch.randn(B, S, 3 * H * D, device="cuda").to(torch.bfloat16)
    q, k, v = qkv[..., :H * D], qkv[..., H * D:2 * H * D], qkv[..., 2 * H * D:]
    out = torch.empty(B, S, H * D, dtype=torch.bfloat16, device="cuda")
    for _ in range(2):
        ops.attention(q, k, v, H, out=out)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for _ in range(iters):
        ops.attention(q, k, v, H, out=out)
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / iters
    fl = 4 * B * H * S * S * D
    print(f"d={D} B={B} H={H} S={S}: {ms * 1e3:.1f} us  {fl / ms / 1e9:.1f} TFLOP/s  ({fl / ms / 1e9 / 2500:.3f} of the bf16 peak)")


if __name__ == "__main__":
    it = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    bench(16, 24, 4224, it)
    bench(16, 24, 4301, it)
    bench(2, 24, 4224, it)
    bench(16, 24, 1152, it)
