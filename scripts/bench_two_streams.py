#!/usr/bin/env python3
"""One transformer forward at CFG batch 16 against two forwards at batch 8 on two streams (would kernel-level concurrency fill
the GEMM tails / epilogue bursts of one half with the other half's kernels?)."""
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
def inputs(B):
    return (torch.randn(B, 16, 64, 64, device=dev).to(torch.bfloat16), torch.full((B,), 700.0, device=dev),
            torch.randn(B, 205, 4096, device=dev).to(torch.bfloat16), torch.randn(B, 2048, device=dev).to(torch.bfloat16))
full, h0, h1 = inputs(16), inputs(8), inputs(8)
s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
def one():
    tr(*full)
def two_serial():
    tr(*h0); tr(*h1)
def two_streams():
    with torch.cuda.stream(s0): tr(*h0)
    with torch.cuda.stream(s1): tr(*h1)
def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
full2 = inputs(16)
def two_full_serial():
    tr(*full); tr(*full2)
def two_full_streams():
    with torch.cuda.stream(s0): tr(*full)
    with torch.cuda.stream(s1): tr(*full2)
print(f"2 x batch 16 serial: {timeit(two_full_serial):.2f} ms; on two streams: {timeit(two_full_streams):.2f} ms")
print(f"batch 16, one stream: {timeit(one):.2f} ms; 2 x batch 8 serial: {timeit(two_serial):.2f} ms; 2 x batch 8 on two streams: {timeit(two_streams):.2f} ms")
