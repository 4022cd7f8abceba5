"""LayerNorm + adaLN-modulate kernel bandwidth at the MMDiT shapes (HBM-bound row kernel)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adv_grpo_amd import ops
def bench(M, D, rows_per_batch, dual, iters=50):
    B = M // rows_per_batch
    x = torch.randn(M, D, device="cuda").to(torch.bfloat16)
    mods = torch.randn(B, 4 * D, device="cuda").to(torch.bfloat16)
    out = torch.empty_like(x)
    kw = dict(scale=mods[:, :D], shift=mods[:, D:2 * D], rows_per_batch=rows_per_batch)
    if dual: kw.update(scale2=mods[:, 2 * D:3 * D], shift2=mods[:, 3 * D:])
    for _ in range(3): ops.layernorm_mod(x, **kw)
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(iters): ops.layernorm_mod(x, **kw)
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / iters * 1e3
    byt = M * D * 2 * (3 if dual else 2)
    print(f"M={M} D={D} dual={dual}: {us:.1f} us  {byt / us / 1e6:.2f} TB/s")
bench(16384, 1536, 1024, False); bench(16384, 1536, 1024, True); bench(3280, 1536, 205, False)
