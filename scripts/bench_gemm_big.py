"""Steady-state GEMM rate on large shapes for one forced variant (ADVGRPO_GEMM_FORCE / ADVGRPO_GEMM_DEBUG)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adv_grpo_amd import ops
def bench(M, N, K, iters=10):
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16); w = torch.randn(N, K, device="cuda").to(torch.bfloat16)
    out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    for _ in range(2): ops.gemm(a, w, out=out)
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(iters): ops.gemm(a, w, out=out)
    e.record(); torch.cuda.synchronize()
    return 2 * M * N * K / (s.elapsed_time(e) / iters) / 1e9
shapes = [(8192, 8192, 8192), (4096, 4096, 32768), (16384, 6144, 1536), (16384, 4608, 1536)]
print("variant", os.environ.get("ADVGRPO_GEMM_FORCE", "auto"), "debug", os.environ.get("ADVGRPO_GEMM_DEBUG", "0"),
      " ".join(f"{bench(*s):7.0f}" for s in shapes))
