import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from adv_grpo_amd import ops
g = torch.Generator(device="cuda").manual_seed(5)
B, H, S, D = 1, 2, 640, 64
q = torch.randn(B, S, H * D, device="cuda", generator=g); k = torch.randn(B, S, H * D, device="cuda", generator=g); v = torch.randn(B, S, H * D, device="cuda", generator=g)
k[:, :64, :] *= 200.0
q, k, v = (x.to(torch.bfloat16) for x in (q, k, v))
lse = torch.empty(B, H, S, dtype=torch.float32, device="cuda")
out = ops.attention(q, k, v, H, lse=lse)
bad = ~torch.isfinite(out.float().view(S, H, D)).all(-1)
print("non-finite (row, head):", bad.nonzero().tolist()[:40], "count", int(bad.sum()))
print("lse nonfinite", (~torch.isfinite(lse)).nonzero().tolist()[:20])
sc = (q.float().view(B, S, H, D).transpose(1, 2) @ k.float().view(B, S, H, D).transpose(1, 2).transpose(-1, -2)) * D ** -0.5 * 1.4427
r, h = bad.nonzero()[0].tolist() if int(bad.sum()) else (0, 0)
print("row", r, "head", h, "tile0 max", sc[0, h, r, :64].max().item(), "overall max", sc[0, h, r].max().item(), "rest max", sc[0, h, r, 64:].max().item())
print(out.float().view(S, H, D)[r, h][:8], lse[0, h, r])
