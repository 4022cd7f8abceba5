"""Yardstick only (NOT part of the product path): torch.matmul (hipBLASLt) next to the eight-phase kernel on the MMDiT GEMM shapes of config 2,
same box, same process, alternating, with the shader clock sampled beside each pair -- so that "0.46 of peak is what this chip gives" is falsifiable
(VERDICT r5 weak 12).  The vendor library computes a plain bf16 GEMM (no bias, no fused epilogue, one weight matrix for all rows); the product kernel
is timed with its bias-only epilogue on one problem of M rows.  M = 16 x 1229 = 19664 rows (CFG batch 16 x (1024 image + 205 text tokens))."""
import os, re, subprocess, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adv_grpo_amd import ops


def clock():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
        c = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", out); p = re.search(r"Power \(W\): ([0-9.]+)", out)
        return f"{c.group(1)} MHz {p.group(1)} W" if c and p else "?"
    except Exception:
        return "?"


def timed(fn, iters=30):
    for _ in range(5): fn()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3          # us


def rnd(*s): return torch.randn(*s, device="cuda").to(torch.bfloat16)


print("shape (M x N x K)            hipBLASLt us  TFLOP/s | gemm8p bias-only us  TFLOP/s | vendor / ours (time) | clock, power after the pair")
for (M, N, K) in [(19664, 1536, 1536), (19664, 4608, 1536), (19664, 6144, 1536), (19664, 1536, 6144), (16384, 4608, 1536), (16384, 6144, 1536),
                  (4096, 4096, 4096), (8192, 8192, 8192)]:
    a, w, b = rnd(M, K), rnd(N, K) * 0.05, rnd(N)
    out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    wt = w.t()
    fl = 2.0 * M * N * K
    rows = []
    for rep in range(2):
        tv = timed(lambda: torch.matmul(a, wt, out=out))
        to = timed(lambda: ops.gemm(a, w, bias=b, out=out))
        rows.append((tv, to, clock()))
    tv, to = min(r[0] for r in rows), min(r[1] for r in rows)
    print(f"{M:6d} x {N:5d} x {K:5d}        {tv:9.1f}  {fl / tv / 1e6:8.0f} | {to:9.1f}          {fl / to / 1e6:8.0f} | {tv / to:5.2f} | {rows[-1][2]}", flush=True)
