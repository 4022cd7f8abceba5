"""Correctness of one forced GEMM tile variant vs an f32 torch matmul (ADVGRPO_GEMM_FORCE=<id>)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adv_grpo_amd import ops
torch.manual_seed(0)
worst = 0.0
for (M, N, K) in [(256, 256, 64), (256, 256, 128), (300, 260, 192), (16384, 1536, 1536), (19664, 512, 320), (512, 384, 1536), (1000, 520, 128), (3280, 1536, 1536)]:
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = torch.randn(N, K, device="cuda").to(torch.bfloat16)
    bias = torch.randn(N, device="cuda").to(torch.bfloat16)
    out = ops.gemm(a, w, bias=bias)
    ref = a.float() @ w.float().t() + bias.float()
    err = ((out.float() - ref).abs().max() / ref.abs().max()).item()
    worst = max(worst, err)
    print(M, N, K, f"rel err {err:.2e}")
assert worst < 1e-2, worst
print("variant", os.environ.get("ADVGRPO_GEMM_FORCE", "auto"), "OK")
