"""Head-dim-80 attention of the CLIP ViT-H vision tower (8 x 16 heads x 257 tokens): time per launch and a hash of the output.
Experiments build: ADVGRPO_ATTN_NO_RESIDENT=1 selects the tiled kernel, default the LDS-resident one; the two must agree bit for bit."""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from adv_grpo_amd import ops  # noqa: E402

torch.manual_seed(0)
for B, H, Sq, Skv in [(8, 16, 257, 257), (16, 16, 257, 257), (2, 16, 50, 50), (1, 4, 300, 320), (3, 2, 512, 65), (2, 3, 1, 257), (2, 16, 577, 577)]:
    g = torch.Generator(device="cuda").manual_seed(Sq * 7 + Skv)
    S = max(Sq, Skv)
    qkv = torch.randn(B, S, 3 * H * 80, device="cuda", generator=g).to(torch.bfloat16)
    q, k, v = qkv[:, :Sq, :H * 80], qkv[:, :Skv, H * 80:2 * H * 80], qkv[:, :Skv, 2 * H * 80:]
    lse = torch.empty(B, H, Sq, dtype=torch.float32, device="cuda")
    out = ops.attention(q, k, v, H, lse=lse)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(50):
        ops.attention(q, k, v, H, out=out)
    e.record()
    torch.cuda.synchronize()
    hsh = hashlib.sha256(out.cpu().view(torch.int16).numpy().tobytes() + lse.cpu().numpy().tobytes()).hexdigest()[:16]
    print(f"B={B} H={H} Sq={Sq} Skv={Skv}: {s.elapsed_time(e) / 50 * 1e3:7.1f} us  sha {hsh}")
