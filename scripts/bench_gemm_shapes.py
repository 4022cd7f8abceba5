"""Times the four grouped Linear launches of one joint block (image + text stream twins) at the rollout's shapes, bf16 and
fp8 operands.  Usage: python scripts/bench_gemm_shapes.py [c2|c4]"""
import sys
import torch

sys.path.insert(0, ".")
from adv_grpo_amd import ops   # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
B, Ni, Nt, D = (16, 1024, 205, 1536) if cfg == "c2" else (8, 4096, 205, 2432)
bf = torch.bfloat16
rnd = lambda *s, k=1.0: (torch.randn(*s, device="cuda") * k).to(bf)
Mi, Mt = B * Ni, B * Nt
shapes = {"qkv": (3 * D, D), "out": (D, D), "ff1": (4 * D, D), "ff2": (D, 4 * D)}


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3


for name, (N, K) in shapes.items():
    xi, xt, wi, wt, bi, bt = rnd(Mi, K), rnd(Mt, K), rnd(N, K, k=0.05), rnd(N, K, k=0.05), rnd(N), rnd(N)
    oi, ot = torch.empty(Mi, N, dtype=bf, device="cuda"), torch.empty(Mt, N, dtype=bf, device="cuda")
    qi, qt, qwi, qwt = (ops.quant_fp8_rows(t) for t in (xi, xt, wi, wt))
    act = "gelu_tanh" if name == "ff1" else None
    f16 = lambda: ops.gemm_grouped([ops.gemm_desc(xi, wi, bias=bi, act=act, out=oi), ops.gemm_desc(xt, wt, bias=bt, act=act, out=ot)])
    f8 = lambda: ops.gemm_grouped_fp8([ops.gemm_desc_fp8(qi, qwi, bias=bi, act=act, out=oi), ops.gemm_desc_fp8(qt, qwt, bias=bt, act=act, out=ot)])
    flops = 2.0 * (Mi + Mt) * N * K
    tiles = (-(-Mi // 256) + -(-Mt // 256)) * -(-N // 256)
    row = [f"{name:4s} N={N:5d} K={K:5d} tiles={tiles:5d} ({tiles / 256:.2f} rounds)"]
    for label, fn in (("bf16", f16), ("fp8", f8)):
        t = timed(fn)
        row.append(f"{label} {t * 1e6:7.1f} us {flops / t / 1e12:7.1f} TF")
    print(" | ".join(row))
