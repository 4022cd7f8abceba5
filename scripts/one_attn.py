import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adv_grpo_amd import ops
B,H,S,D=16,24,1229,64
qkv=torch.randn(B,S,3*H*D,device='cuda').to(torch.bfloat16)
q,k,v=qkv[...,:H*D],qkv[...,H*D:2*H*D],qkv[...,2*H*D:]
out=torch.empty(B,S,H*D,dtype=torch.bfloat16,device='cuda')
for _ in range(5): ops.attention(q,k,v,H,out=out)
torch.cuda.synchronize()
