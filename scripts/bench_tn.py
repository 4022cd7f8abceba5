"""Token-contracted GEMM (LoRA weight gradient) timing: ADVGRPO_TN_BLOCKS=<target workgroups> python scripts/bench_tn.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adv_grpo_amd import ops
def bench(M, N1, iters=20):
    P = torch.randn(M, N1, device="cuda").to(torch.bfloat16); Q = torch.randn(M, 64, device="cuda").to(torch.bfloat16)
    out = torch.zeros(N1, 64, dtype=torch.float32, device="cuda")
    for _ in range(3): ops.gemm_tn(P, Q, out)
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(iters): ops.gemm_tn(P, Q, out)
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / iters * 1e3
    return us, (M * N1 * 2 + M * 128) / us / 1e6
for M in (16384, 3280):
    us, tbs = bench(M, 1536)
    print(f"blocks {os.environ.get('ADVGRPO_TN_BLOCKS', '1024')} M={M}: {us:.1f} us  {tbs:.2f} TB/s")
