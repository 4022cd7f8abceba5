import sys, torch
sys.path.insert(0, ".")
from adv_grpo_amd import ops
torch.manual_seed(0)
for (M, N, K) in [(16384, 1536, 1536), (19664, 4608, 1536), (20000, 1024, 512)]:
    x = torch.randn(M, K, device="cuda").to(torch.bfloat16); w = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
    with ops.stream_k(False):
        ref = ops.gemm(x, w)
    for rep in range(3):
        with ops.stream_k(True):
            out = ops.gemm(x, w)
        torch.cuda.synchronize()
        d = (out.float() - ref.float()).abs()
        bad = d > (ref.float().abs() * 2 ** -6 + 1e-2)
        rows = bad.any(dim=1).nonzero().flatten()
        cols = bad.any(dim=0).nonzero().flatten()
        print(M, N, K, "rep", rep, "max diff", d.max().item(), "bad elements", bad.sum().item(),
              "row tiles", sorted(set((rows // 256).tolist()))[:12], "col tiles", sorted(set((cols // 256).tolist()))[:12])
