import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from adv_grpo_amd import ops
for (B, H, S) in [(8, 4, 77), (8, 4, 64)]:
    torch.manual_seed(S + B)
    qkv = torch.randn(B, S, 3 * H * 64, device='cuda').to(torch.bfloat16)
    q, k, v = qkv[..., :H * 64], qkv[..., H * 64:2 * H * 64], qkv[..., 2 * H * 64:]
    outs, lses = [], []
    for i in range(4):
        lse = torch.empty(B, H, S, dtype=torch.float32, device='cuda')
        outs.append(ops.attention(q, k, v, H, lse=lse).clone()); lses.append(lse)
    torch.cuda.synchronize()
    for i in range(1, 4):
        d = (outs[0] != outs[i])
        dl = (lses[0] != lses[i])
        rows = d.view(B, S, H, 64).any(-1)          # [B,S,H]
        print((B, H, S), "run", i, "diff elems", int(d.sum()), "diff (b,h) pairs:", sorted(set((int(b), int(h)) for b, s, h in rows.nonzero().tolist()))[:12],
              "rows per pair", int(rows.sum()) , "lse diffs", int(dl.sum()), "max lse diff", (lses[0] - lses[i]).abs().max().item())
        if int(d.sum()):
            b, s, h = rows.nonzero()[0].tolist()
            print("   first differing row", (b, s, h), "d cols differing", d.view(B, S, H, 64)[b, s, h].nonzero().flatten().tolist()[:20])
            print("   rows differing in that (b,h):", rows[b, :, h].nonzero().flatten().tolist()[:40])
