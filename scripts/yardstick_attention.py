"""Yardstick only (NOT part of the product path): torch's F.scaled_dot_product_attention (the vendor flash-attention kernel PyTorch-ROCm dispatches to)
next to attention_fwd_pipe_kernel at the MMDiT shapes, same box, same process, alternating.  (VERDICT r5 asked for the guide's tuned loop
`examples/attn_fwd_pwg4x64_bf16.cpp` at 16 x 24 x 1229: /opt/skills/guides/ holds the two .md files only in this image -- no examples/ directory -- so the
vendor kernel stands in; the guide's own figures for that loop are ~1200 TFLOP/s at head dim 128, S = 2048, random data.)  Head dim 64, bf16, no mask."""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adv_grpo_amd import ops


def timed(fn, iters=50, reps=5):
    for _ in range(10): fn()
    best = 1e9
    for _ in range(reps):
        s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); s.record()
        for _ in range(iters): fn()
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / iters * 1e3)
    return best


print("B x H x S (d = 64)      vendor SDPA us  TFLOP/s | attention_fwd_pipe us  TFLOP/s | max |diff| of the two outputs")
for (B, H, S) in [(16, 24, 1229), (16, 24, 1024), (8, 24, 1229), (16, 24, 4301)]:
    D = 64
    qkv = torch.randn(B, S, 3 * H * D, device="cuda").to(torch.bfloat16)
    q, k, v = qkv[..., :H * D], qkv[..., H * D:2 * H * D], qkv[..., 2 * H * D:]
    out = torch.empty(B, S, H * D, dtype=torch.bfloat16, device="cuda")
    q4, k4, v4 = (t.reshape(B, S, H, D).transpose(1, 2).contiguous() for t in (q, k, v))        # [B, H, S, D], contiguous: the vendor kernel's best case
    fl = 4.0 * B * H * S * S * D
    backends = []
    try:
        from torch.nn.attention import SDPBackend, sdpa_kernel
        ctx = lambda: sdpa_kernel([SDPBackend.FLASH_ATTENTION, SDPBackend.EFFICIENT_ATTENTION])
    except Exception:
        import contextlib
        ctx = contextlib.nullcontext
    try:
        with ctx():
            ref = F.scaled_dot_product_attention(q4, k4, v4)
            tv = timed(lambda: F.scaled_dot_product_attention(q4, k4, v4))
    except Exception as ex:            # noqa: BLE001
        ref, tv = None, float("nan")
        print("vendor SDPA failed:", type(ex).__name__, str(ex)[:200])
    to = timed(lambda: ops.attention(q, k, v, H, out=out))
    diff = (ref.transpose(1, 2).reshape(B, S, H * D).float() - out.float()).abs().max().item() if ref is not None else float("nan")
    print(f"{B:2d} x {H} x {S:5d}        {tv:9.1f}  {fl / tv / 1e6:8.0f} | {to:9.1f}           {fl / to / 1e6:8.0f} | {diff:.4f}", flush=True)
