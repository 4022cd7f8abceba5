"""A few launches of the attention forward at one shape (for rocprofv3 --pmc / --kernel-trace runs).  Usage: one_attn.py [S] [B] [H]"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adv_grpo_amd import ops
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1229
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
H = int(sys.argv[3]) if len(sys.argv) > 3 else 24
D = 64
qkv = torch.randn(B, S, 3 * H * D, device='cuda').to(torch.bfloat16)
q, k, v = qkv[..., :H * D], qkv[..., H * D:2 * H * D], qkv[..., 2 * H * D:]
out = torch.empty(B, S, H * D, dtype=torch.bfloat16, device='cuda')
for _ in range(5):
    ops.attention(q, k, v, H, out=out)
torch.cuda.synchronize()
