"""Is the 8-phase kernel's epilogue bandwidth- or latency-bound?  One round of T tiles (T CUs busy), K fixed: the per-tile
time must not depend on T if the epilogue is latency-bound.  ADVGRPO_GEMM_FORCE=30."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adv_grpo_amd import ops
def t_us(M, N, K, epi, iters=20):
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16); w = torch.randn(N, K, device="cuda").to(torch.bfloat16)
    out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda"); res = torch.randn(M, N, device="cuda").to(torch.bfloat16)
    kw = {}
    if epi: kw = dict(bias=torch.randn(N, device="cuda").to(torch.bfloat16), gate=torch.randn(16, N, device="cuda").to(torch.bfloat16), gate_rows=(M + 15) // 16, residual=res)
    for _ in range(3): ops.gemm(a, w, out=out, **kw)
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(iters): ops.gemm(a, w, out=out, **kw)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
for K in (1536, 3072):
    for T in (8, 32, 64, 128, 256):
        print(f"K={K} tiles={T:4d}: plain {t_us(256 * T // 4, 1024, K, False):7.1f} us   bias+gate+residual {t_us(256 * T // 4, 1024, K, True):7.1f} us")
