"""Same-box A/B of the paired LayerNorm launch (mmdit.pair_norms): one CFG forward of SD3.5-medium at 512^2, batch 16."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from adv_grpo_amd import mmdit, synthetic  # noqa: E402
from adv_grpo_amd.model_configs import MMDiTConfig  # noqa: E402

dev = torch.device("cuda", 0)
cfg = MMDiTConfig()
with synthetic.on_device(dev):
    tr = mmdit.SD3Transformer2DModel(synthetic.mmdit_weights(cfg, 1), cfg, dev)
B = 16
x = torch.randn(B, 16, 64, 64, device=dev).to(torch.bfloat16)
t = torch.full((B,), 500.0, device=dev)
ctx = torch.randn(B, 154, 4096, device=dev).to(torch.bfloat16)
pooled = torch.randn(B, 2048, device=dev).to(torch.bfloat16)


def run(n):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        y = tr(x, t, ctx, pooled)[0]
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n, y


outs = {}
for rep in range(3):
    for flag in (True, False):
        tr.pair_norms = flag
        run(2)
        ms, y = run(10)
        outs[flag] = y.clone()
        print(f"pair_norms={flag}: {ms:.3f} ms per forward")
print("bit-identical:", torch.equal(outs[True], outs[False]))
