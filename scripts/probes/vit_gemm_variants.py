"""The four Linears of a CLIP ViT-H layer at the scorer's M (8 images x 257 tokens = 2056 rows) under one tile variant
(ADVGRPO_GEMM_FORCE, experiments build; 'auto' = the product dispatch).  One process per variant: the knob is read once."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from adv_grpo_amd import ops  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 2056
D, F = 1280, 5120
bf = torch.bfloat16
rnd = lambda *s, k=1.0: (torch.randn(*s, device="cuda") * k).to(bf)
shapes = [("qkv", 3 * D, D, None, False), ("out", D, D, None, True), ("fc1", F, D, "gelu", False), ("fc2", D, F, None, True)]


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3


tot = 0.0
row = [f"FORCE={os.environ.get('ADVGRPO_GEMM_FORCE', 'auto'):>4s} M={M}"]
for name, N, K, act, res in shapes:
    x, w, b = rnd(M, K), rnd(N, K, k=0.03), rnd(N)
    out = torch.empty(M, N, dtype=bf, device="cuda")
    r = rnd(M, N) if res else None
    try:
        fn = lambda: ops.gemm(x, w, bias=b, act=act, residual=r, out=out)
        t = timed(fn)
    except Exception as ex:   # a variant that refuses the shape
        row.append(f"{name} refused ({str(ex)[:40]})")
        continue
    tot += t
    row.append(f"{name} {t * 1e6:6.1f} us {2.0 * M * N * K / t / 1e12:6.0f} TF")
row.append(f"layer {tot * 1e6:6.1f} us")
print(" | ".join(row))
