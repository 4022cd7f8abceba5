#!/usr/bin/env python3
"""How much of a transformer forward is launch gaps: 10 eager forwards at the config-2 CFG batch against 10 replays of ONE
captured hipGraph of the same forward (torch.cuda.graph; the ctypes-launched kernels run on torch's capture stream)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adv_grpo_amd import synthetic
from adv_grpo_amd.mmdit import SD3Transformer2DModel
from adv_grpo_amd.model_configs import MMDiTConfig

dev = "cuda"
cfg = MMDiTConfig()
with synthetic.on_device(dev):
    tr = SD3Transformer2DModel(synthetic.mmdit_weights(cfg, 1234), cfg, dev)
x = torch.randn(16, 16, 64, 64, device=dev).to(torch.bfloat16)
t = torch.full((16,), 700.0, device=dev)
ctx = torch.randn(16, 205, 4096, device=dev).to(torch.bfloat16)
pooled = torch.randn(16, 2048, device=dev).to(torch.bfloat16)
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
eager = timeit(lambda: tr(x, t, ctx, pooled))
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(2): tr(x, t, ctx, pooled)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s):
    out = tr(x, t, ctx, pooled)[0]
graph = timeit(g.replay)
ref = tr(x, t, ctx, pooled)[0]
g.replay(); torch.cuda.synchronize()
print(f"eager {eager:.2f} ms  graph replay {graph:.2f} ms  ({100 * (eager - graph) / eager:.1f} % saved)  identical: {torch.equal(out, ref)}")
