"""One shape through the forced GEMM variant a few times (for rocprofv3 --pmc passes)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adv_grpo_amd import ops
M, N, K = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (4096, 4096, 8192)))
a = torch.randn(M, K, device="cuda").to(torch.bfloat16); w = torch.randn(N, K, device="cuda").to(torch.bfloat16)
out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
for _ in range(4): ops.gemm(a, w, out=out)
torch.cuda.synchronize()
