#!/usr/bin/env python3
"""GroupNorm + SiLU -> split rows of the fp32-equivalent VAE decode at the decoder's four resolutions (8 images)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adv_grpo_amd import ops

def timeit(fn, iters=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

for hw, C in ((512, 128), (256, 256), (128, 512), (64, 512)):
    x = torch.randn(8, hw, hw, C, device="cuda")
    w, b = torch.randn(C, device="cuda"), torch.randn(C, device="cuda")
    for po in (False, True):
        us = timeit(lambda: ops.groupnorm_nhwc_x3(x, w, b, 32, 1e-6, True, po))
        byt = x.numel() * (4 + 4 + (4 if po else 6))
        print(f"8 x {hw}^2 x {C} pair_only={int(po)}: {us:8.1f} us  {byt / us / 1e6:5.2f} TB/s (stats read + apply read + write)")
    us = timeit(lambda: ops.split_x3(x, 2))
    print(f"8 x {hw}^2 x {C} split order 2: {us:8.1f} us  {x.numel() * 8 / us / 1e6:5.2f} TB/s")
