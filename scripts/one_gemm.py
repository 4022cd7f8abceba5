import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adv_grpo_amd import ops
M,N,K = [int(x) for x in sys.argv[1:4]]
a=torch.randn(M,K,device='cuda').to(torch.bfloat16); w=torch.randn(N,K,device='cuda').to(torch.bfloat16)
out=torch.empty(M,N,dtype=torch.bfloat16,device='cuda')
for _ in range(5): ops.gemm(a,w,out=out)
torch.cuda.synchronize()
